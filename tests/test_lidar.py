"""LiDAR front end (BASELINE config 5).  CPU: the oracle restatement against vectors produced by the reference's own
range_finder / scikit-learn / Qhull (tests/golden/lidar_golden.npz).  GPU: lipmpc_lidar_sense_batch against both."""
import os

import numpy as np
import pytest

import lidar_oracle as L


def _case(d, i):
    rings = [d["env"][i][j][: d["env_nv"][i][j]] for j in range(d["env"].shape[1]) if d["env_nv"][i][j] > 0]
    return d["pos"][i], rings, float(d["lidar_range"][i])


def _same_ring(a, b):
    if len(a) != len(b):
        return False
    k = int(np.argmin(np.abs(b - a[0]).sum(1)))
    return np.array_equal(np.roll(b, -k, axis=0), a)


def test_oracle_matches_reference_range_finder(golden_dir):
    d = np.load(os.path.join(golden_dir, "lidar_golden.npz"))
    tab = L.ray_table()
    for i in range(0, len(d["pos"]), 3):
        pos, rings, rng = _case(d, i)
        hits, valid = L.lidar_hits(pos, rings, rng, tab)
        assert np.array_equal(valid, d["valid"][i])
        assert np.array_equal(hits[valid], d["clean"][i][valid])                 # hit points bit-exact
        _, _, labels, inferred = L.range_finder(pos, rings, rng, noise=d["noise"][i], table=tab)
        assert np.array_equal(labels, d["labels"][i][valid])                      # scikit-learn's DBSCAN labels
        n_ref = int((d["inf_nv"][i] > 0).sum())
        assert len(inferred) == n_ref
        for j, ring in enumerate(inferred):                                        # Qhull's rings (any rotation)
            assert _same_ring(ring, d["inf_xy"][i][j][: d["inf_nv"][i][j]])


def _res_case(d, i):
    R = int(d["res"][i])
    rings = [d["env"][i][j][: d["env_nv"][i][j]] for j in range(d["env"].shape[1]) if d["env_nv"][i][j] > 0]
    return d["pos"][i], rings, float(d["lidar_range"][i]), R


def test_oracle_matches_reference_range_finder_at_other_resolutions(golden_dir):
    """lidar_golden_res.npz: the reference's range_finder at 90 / 180 / 270 rays and 1 / 2 m (made by importing it,
    tests/golden/make_lidar_golden_res.py): hit points bit-exact, scikit-learn's labels, Qhull's rings."""
    d = np.load(os.path.join(golden_dir, "lidar_golden_res.npz"))
    for i in range(len(d["pos"])):
        pos, rings, rng, R = _res_case(d, i)
        tab = L.ray_table(R)
        hits, valid = L.lidar_hits(pos, rings, rng, tab)
        assert np.array_equal(valid, d["valid"][i][:R])
        assert np.array_equal(hits[valid], d["clean"][i][:R][valid])
        _, _, labels, inferred = L.range_finder(pos, rings, rng, noise=d["noise"][i][:R], table=tab)
        assert np.array_equal(labels, d["labels"][i][:R][valid])
        assert len(inferred) == int((d["inf_nv"][i] > 0).sum())
        for j, ring in enumerate(inferred):
            assert _same_ring(ring, d["inf_xy"][i][j][: d["inf_nv"][i][j]])


@pytest.mark.gpu
def test_gpu_lidar_matches_reference_at_other_resolutions(golden_dir):
    """The kernel against the same reference-made vectors: 90 / 180 / 270 rays, ranges of 1 and 2 m."""
    torch = pytest.importorskip("torch")
    import lipmpc
    d = np.load(os.path.join(golden_dir, "lidar_golden_res.npz"))
    for i in range(len(d["pos"])):
        pos, rings, rng, R = _res_case(d, i)
        sensor = lipmpc.LidarSensor(rings, lidar_range=rng, resolution=R, n_obs_max=12, v_max=40)
        st = torch.tensor([[pos[0], 0.0, pos[1], 0.0, 0.0]], dtype=torch.float64, device="cuda")
        noise = torch.as_tensor(np.ascontiguousarray(d["noise"][i][None, :R]), device="cuda")
        out = sensor.sense(st, noise, with_debug=True)
        torch.cuda.synchronize()
        hits = out["hits"][0].cpu().numpy()
        valid = ~np.isnan(hits[:, 0])
        assert np.array_equal(valid, d["valid"][i][:R])
        assert np.array_equal(hits[valid], (d["clean"][i] + d["noise"][i])[:R][valid])
        assert np.array_equal(out["labels"][0].cpu().numpy(), d["labels"][i][:R])
        n = int(out["n_inferred"][0]); nv = out["obs_nv"][0].cpu().numpy(); xy = out["obs_xy"][0].cpu().numpy()
        assert int(out["overflow"][0]) == 0 and n == int((d["inf_nv"][i] > 0).sum())
        for j in range(n):
            assert _same_ring(xy[j, : nv[j]], d["inf_xy"][i][j][: d["inf_nv"][i][j]]), (i, j)


def test_dbscan_border_and_noise_rules():
    # two 3-point cores 0.5 apart with one border point within eps of both: it joins the first-discovered cluster
    pts = np.array([[0.0, 0.0], [0.1, 0.0], [0.2, 0.0], [0.45, 0.0], [0.7, 0.0], [0.8, 0.0], [0.9, 0.0], [3.0, 3.0]])
    from sklearn.cluster import DBSCAN
    ref = DBSCAN(eps=0.3, min_samples=3).fit(pts).labels_
    assert np.array_equal(L.dbscan_labels(pts), ref)
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = rng.uniform(0, 3, (int(rng.integers(5, 120)), 2))
        assert np.array_equal(L.dbscan_labels(p), DBSCAN(eps=0.3, min_samples=3).fit(p).labels_)
    # other eps / min_samples (the settings test_gpu_clustering_routes_against_oracle runs the kernel at)
    for eps, ms in ((0.3, 2), (0.3, 4), (0.3, 5), (0.3, 1), (0.3, 7), (0.12, 3), (0.6, 3), (0.04, 3), (0.02, 2)):
        for _ in range(6):
            p = rng.uniform(0, 3 * eps / 0.3, (int(rng.integers(5, 120)), 2))
            assert np.array_equal(L.dbscan_labels(p, eps, ms), DBSCAN(eps=eps, min_samples=ms).fit(p).labels_), (eps, ms)


def test_hull_rules():
    assert L.hull_ring(np.array([[0.0, 0], [1, 1], [2, 2], [3, 3]])) is None        # collinear (rank test, :75)
    assert L.hull_ring(np.array([[0.0, 0], [1, 1], [0, 0]])) is None                # < 3 unique points (:70)
    r = L.hull_ring(np.array([[0.0, 0], [1, 0], [2, 0], [2, 2], [0, 2], [1, 1], [1, 0]]))
    assert np.array_equal(r, np.array([[0.0, 0], [2, 0], [2, 2], [0, 2]]))          # CCW, edge midpoint dropped


@pytest.mark.parametrize("noise_seed", [0, 3])
def test_unknown_environment_closed_loop_against_the_reference_figure(golden_dir, noise_seed):
    """BASELINE config 5 end to end against the reference's own output: the unknown-environment run the reference commits
    (Assets/ReportResults/Simulation4UnkEnv, produced by simulation_1.py:195-232: CROWDED map under seed 10, start
    (0, 0, pi/2), goal (4, 3.5), N = 3, lidar_range 1.5) on the oracle chain lidar oracle -> step oracle.  The reference's
    sensor noise was unseeded, so the figure pins the loop as far as sigma = 0.01 m of noise lets it -- which is: the same
    run length (57 states), the first steps to 1e-7, ten steps to 0.5 mm, the whole run to 4 cm, for any noise sample."""
    from helpers import check_pdf_bars, oracle_unknown_env_run, pdf_compare
    X, U = oracle_unknown_env_run(golden_dir, noise_seed)
    cmp = pdf_compare(golden_dir, "Simulation4UnkEnv", X, U)
    print("Simulation4UnkEnv, oracle chain, noise seed", noise_seed, X.shape[1], cmp)
    check_pdf_bars("Simulation4UnkEnv", X, cmp)
    assert np.hypot(X[0, -1] - 4.0, X[2, -1] - 3.5) < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("noise_seed", [0, 1])
def test_gpu_unknown_environment_class_against_the_reference_figure(golden_dir, noise_seed):
    """The same pin through the GPU drop-in class lipmpc.HumanoidMPCUnknownEnvironment (scan + constraint assembly in one
    launch per sample, then the solve): the reference's committed unknown-environment figure, same bars as the oracle chain."""
    pytest.importorskip("torch")
    import lipmpc
    from helpers import check_pdf_bars, pdf_compare, unknown_env_scenario
    sc = unknown_env_scenario(golden_dir)
    mpc = lipmpc.HumanoidMPCUnknownEnvironment(goal=sc["goal"], obstacles=sc["env"], N_horizon=sc["N"], N_mpc_timesteps=300,
                                               sampling_time=0.4, init_state=sc["init"], verbosity=0,
                                               lidar_range=sc["lidar_range"], noise_seed=noise_seed)
    X, U, _ = mpc.run_simulation(None, make_fast_plot=False, fill_animator=False)
    cmp = pdf_compare(golden_dir, "Simulation4UnkEnv", X, U)
    print("Simulation4UnkEnv, GPU class, noise seed", noise_seed, X.shape[1], cmp)
    check_pdf_bars("Simulation4UnkEnv", X, cmp)
    assert np.hypot(X[0, -1] - 4.0, X[2, -1] - 3.5) < 0.1


@pytest.mark.gpu
def test_gpu_lidar_matches_reference_and_oracle(golden_dir):
    torch = pytest.importorskip("torch")
    import lipmpc
    d = np.load(os.path.join(golden_dir, "lidar_golden.npz"))
    n_bad = 0
    for i in range(len(d["pos"])):
        pos, rings, rng = _case(d, i)
        sensor = lipmpc.LidarSensor(rings, lidar_range=rng, resolution=360, n_obs_max=12, v_max=40)
        st = torch.tensor([[pos[0], 0.0, pos[1], 0.0, 0.0]], dtype=torch.float64, device="cuda")
        noise = torch.as_tensor(d["noise"][i][None], device="cuda")
        out = sensor.sense(st, noise, with_debug=True)
        torch.cuda.synchronize()
        hits = out["hits"][0].cpu().numpy()
        valid = ~np.isnan(hits[:, 0])
        assert np.array_equal(valid, d["valid"][i])
        assert np.array_equal(hits[valid], (d["clean"][i] + d["noise"][i])[valid])        # bit-exact readings
        assert np.array_equal(out["labels"][0].cpu().numpy(), d["labels"][i])              # scikit-learn's labels
        n = int(out["n_inferred"][0]); nv = out["obs_nv"][0].cpu().numpy(); xy = out["obs_xy"][0].cpu().numpy()
        assert int(out["overflow"][0]) == 0
        assert n == int((d["inf_nv"][i] > 0).sum())
        for j in range(n):
            ref = d["inf_xy"][i][j][: d["inf_nv"][i][j]]
            assert _same_ring(xy[j, : nv[j]], ref), (i, j)
    assert n_bad == 0


@pytest.mark.gpu
def test_gpu_unknown_environment_step_end_to_end(golden_dir):
    """scan -> rings -> plan_step on the device against lidar oracle -> step oracle, many robots on one shared map."""
    torch = pytest.importorskip("torch")
    import lipmpc
    import lipmpc_oracle as O
    d = np.load(os.path.join(golden_dir, "lidar_golden.npz"))
    _, rings, _ = _case(d, 0)
    rng = np.random.default_rng(3)
    B, N = 96, 3
    pos = []
    while len(pos) < B:
        p = rng.uniform(-0.5, 5.5, 2)
        if not any(O.point_in_ring(p, r) for r in rings):
            pos.append(p)
    pos = np.array(pos)
    st = np.zeros((B, 5)); st[:, 0] = pos[:, 0]; st[:, 2] = pos[:, 1]
    noise = 0.01 * rng.standard_normal((B, 360, 2))
    sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, n_obs_max=12, v_max=32)
    d_st = torch.as_tensor(st, device="cuda")
    sen = sensor.sense(d_st, torch.as_tensor(noise, device="cuda"))
    P = lipmpc.LipMpcParams(N=N, n_obs_max=12, v_max=32)
    sv = lipmpc.BatchedLipMpc(P)
    goal = torch.tensor([[5.0, 5.0]], dtype=torch.float64, device="cuda").repeat(B, 1).contiguous()
    foot = torch.ones((B,), dtype=torch.int8, device="cuda")
    out = sv.plan_step_batch(d_st, goal, foot, sen["obs_xy"], sen["obs_nv"], None)
    torch.cuda.synchronize()
    assert int(sen["overflow"].sum()) == 0
    U, status = out["U"].cpu().numpy(), out["status"].cpu().numpy()
    tab = L.ray_table()
    n_ok = 0
    for b in range(B):
        _, _, _, inferred = L.range_finder(pos[b], rings, 1.5, noise=noise[b], table=tab)
        r = O.plan_step(st[b], (5.0, 5.0), 1, inferred, 0.0, O.Params(N=N))
        assert int(sen["n_inferred"][b]) == len(inferred)
        assert status[b] == r["status"], (b, status[b], r["status"])
        if r["status"] == 0:
            n_ok += 1
            assert np.max(np.abs(U[b] - r["U"])) < 1e-7
    assert n_ok > 0.7 * B


@pytest.mark.gpu
def test_gpu_unknown_environment_class_between_mpc_samples(golden_dir, monkeypatch):
    """sampling_time = 0.1 (mpc_step = 4): like the reference, the class scans on EVERY sample (its lists grow per sample,
    HumanoidMpc.py:387 / HumanoidMPCUnknownEnvironment.py:65-66) but solves only on MPC samples (:415-417), between which
    the CoM stands still and only the heading moves (:443-447); list_lidar_readings mirrors range_finder's readings
    (`resolution` entries, None or the noisy hit)."""
    torch = pytest.importorskip("torch")
    import lipmpc
    d = np.load(os.path.join(golden_dir, "lidar_golden.npz"))
    _, rings, _ = _case(d, 0)
    mpc = lipmpc.HumanoidMPCUnknownEnvironment(goal=(5, 5), obstacles=rings, N_horizon=3, N_mpc_timesteps=4, sampling_time=0.1,
                                               init_state=(-0.8, 0, -0.8, 0, 0.7), verbosity=0, lidar_range=1.5, noise_seed=1)
    solves = []
    plan = lipmpc.HumanoidMPC._plan
    monkeypatch.setattr(lipmpc.HumanoidMPC, "_plan", lambda self, st, s0, lists=None: (solves.append(1), plan(self, st, s0, lists))[1])
    X, U, _ = mpc.run_simulation(None, make_fast_plot=False, fill_animator=False)
    K = U.shape[1]
    assert K == 15 and X.shape[1] == 16                       # 4 MPC steps x 4 samples, the reference's truncation drops the last
    assert len(solves) == 4                                    # one solve per MPC sample, none in between
    assert len(mpc.list_inferred_obstacles) == 16 and len(mpc.list_lidar_readings) == 16     # one scan per sample
    for k in range(K):
        if k % 4 != 0:                                         # between MPC samples: CoM state kept, footstep re-applied
            assert np.array_equal(X[:4, k + 1], X[:4, k]) and np.array_equal(U[:2, k], U[:2, k - 1])
        else:
            assert not np.array_equal(X[:4, k + 1], X[:4, k])
    assert np.all(np.diff(X[4]) != 0.0)                         # the heading moves on every sample
    rd = mpc.list_lidar_readings[0]
    assert len(rd) == 360 and any(r is None for r in rd) and any(r is not None for r in rd)
    pts = np.array([r for r in rd if r is not None])
    assert pts.shape[1] == 2 and np.all(np.hypot(pts[:, 0] + 0.8, pts[:, 1] + 0.8) < 1.5 + 0.1)     # hits within the LiDAR range of the robot
    # A run that stops on the objective: the reference assembles the constraints (the subclass hook scans) BEFORE its stop test
    # (HumanoidMpc.py:387 vs :392), so the sample it stops at is scanned too: one list entry per kept state
    near = lipmpc.HumanoidMPCUnknownEnvironment(goal=(-0.7, -0.72), obstacles=rings, N_horizon=3, N_mpc_timesteps=30, sampling_time=0.4,
                                                init_state=(-0.8, 0, -0.8, 0, 0.7), verbosity=0, lidar_range=1.5, noise_seed=1)
    Xn, Un, _ = near.run_simulation(None, make_fast_plot=False, fill_animator=False)
    assert 2 <= Xn.shape[1] < 30 and near.last_status in (0, 4)                 # stopped early, on the objective
    assert len(near.list_inferred_obstacles) == Xn.shape[1] == len(near.list_lidar_readings)


@pytest.mark.gpu
def test_gpu_unknown_environment_class_and_throughput(golden_dir):
    """The drop-in HumanoidMPCUnknownEnvironment walks a CROWDED map it only sees through the scanner, and the
    batched pipeline (scan + step for 4096 robots on one map) is timed for the record."""
    torch = pytest.importorskip("torch")
    import lipmpc
    import lipmpc_oracle as O
    d = np.load(os.path.join(golden_dir, "lidar_golden.npz"))
    _, rings, _ = _case(d, 0)
    mpc = lipmpc.HumanoidMPCUnknownEnvironment(goal=(5, 5), obstacles=rings, N_horizon=3, N_mpc_timesteps=60,
                                               sampling_time=0.4, init_state=(-0.8, 0, -0.8, 0, 0.7), verbosity=0,
                                               lidar_range=1.5, noise_seed=1)
    X, U, _ = mpc.run_simulation(None, make_fast_plot=False, fill_animator=False)
    assert X.shape[1] >= 10 and len(mpc.list_inferred_obstacles) >= 10
    for k in range(X.shape[1]):                                  # never inside a true obstacle
        assert not any(O.point_in_ring((X[0, k], X[2, k]), r) for r in rings)
    assert np.hypot(X[0, -1] - 5, X[2, -1] - 5) < np.hypot(X[0, 0] - 5, X[2, 0] - 5)   # and it made progress
    # batched: 4096 robots on the shared map
    B = 4096
    rng = np.random.default_rng(0)
    pos = rng.uniform(-0.8, 5.8, (B, 2))
    st = np.zeros((B, 5)); st[:, 0] = pos[:, 0]; st[:, 2] = pos[:, 1]
    d_st = torch.as_tensor(st, device="cuda")
    sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, n_obs_max=12, v_max=32)
    noise = 0.01 * torch.randn((B, 360, 2), dtype=torch.float64, device="cuda")
    sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=3, n_obs_max=12, v_max=32))
    goal = torch.tensor([[5.0, 5.0]], dtype=torch.float64, device="cuda").repeat(B, 1).contiguous()
    foot = torch.ones((B,), dtype=torch.int8, device="cuda")
    for _ in range(2):
        sen = sensor.sense(d_st, noise); out = sv.plan_step_batch(d_st, goal, foot, sen["obs_xy"], sen["obs_nv"], None)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); sen = sensor.sense(d_st, noise); e[1].record()
    out = sv.plan_step_batch(d_st, goal, foot, sen["obs_xy"], sen["obs_nv"], None); e[2].record()
    torch.cuda.synchronize()
    print(f"config 5, B={B}: scan {e[0].elapsed_time(e[1]):.3f} ms + step {e[1].elapsed_time(e[2]):.3f} ms; "
          f"inferred obstacles mean {float(sen['n_inferred'].double().mean()):.2f}, overflow {int(sen['overflow'].sum())}")
    assert int(sen["overflow"].sum()) == 0


@pytest.mark.gpu
def test_gpu_unknown_environment_fleet_matches_class(golden_dir):
    """UnknownEnvFleet (scan + solve + advance enqueued per sample, captured in a HIP graph) against the drop-in
    class driven sample by sample from the host, on the same noise stream: identical trajectories; robots with
    other noise / starts walk the same map in the same launches."""
    torch = pytest.importorskip("torch")
    import lipmpc
    d = np.load(os.path.join(golden_dir, "lidar_golden.npz"))
    _, rings, _ = _case(d, 0)
    K = 25
    mpc = lipmpc.HumanoidMPCUnknownEnvironment(goal=(5, 5), obstacles=rings, N_horizon=3, N_mpc_timesteps=K,
                                               sampling_time=0.4, init_state=(-0.8, 0, -0.8, 0, 0.7), verbosity=0,
                                               lidar_range=1.5, noise_seed=7)
    X, U, _ = mpc.run_simulation(None, make_fast_plot=False, fill_animator=False)
    # the class draws randn((1, 360, 2)) per sample from a generator seeded 7: rebuild that stream for robot 0
    gen = torch.Generator(device="cuda").manual_seed(7)
    B = 4
    noise = torch.zeros((K, B, 360, 2), dtype=torch.float64, device="cuda")
    for k in range(K):
        noise[k, 0] = 0.01 * torch.randn((1, 360, 2), dtype=torch.float64, device="cuda", generator=gen)[0]
    noise[:, 1:] = 0.01 * torch.randn((K, B - 1, 360, 2), dtype=torch.float64, device="cuda")
    st0 = torch.tensor([[-0.8, 0, -0.8, 0, 0.7]] * B, dtype=torch.float64, device="cuda")
    st0[3, 0] = 5.5; st0[3, 2] = -0.5; st0[3, 4] = 2.0
    goal = torch.tensor([[5.0, 5.0]] * B, dtype=torch.float64, device="cuda")
    foot = torch.ones((B,), dtype=torch.int8, device="cuda")
    res = {}
    for use_graph in (False, True):
        fleet = lipmpc.UnknownEnvFleet(rings, N_horizon=3, lidar_range=1.5)
        r = fleet.run(st0, goal, foot, K, noise=noise, use_graph=use_graph)
        torch.cuda.synchronize()
        res[use_graph] = {k: v.cpu().numpy() for k, v in r.items()}
    for use_graph in (False, True):
        r = res[use_graph]
        n0 = int(r["n_steps"][0])
        kk = n0 if n0 < K else K - 1                    # the class's truncation (HumanoidMpc.py:457-459)
        assert X.shape[1] == kk + 1, (X.shape, n0)
        assert np.max(np.abs(r["X_pred"][0, :kk + 1].T - X)) < 1e-9
        assert np.max(np.abs(r["U_pred"][0, :kk].T - U)) < 1e-9
        assert r["n_steps"].min() >= 5 and r["overflow"].sum() == 0
    assert np.array_equal(res[False]["X_pred"], res[True]["X_pred"])
    assert np.array_equal(res[False]["n_steps"], res[True]["n_steps"])
    # seeded noise drawn INSIDE the captured graph: a run is a function of its seed (the second run of a shape replays
    # the graph the first one captured), and another seed walks another way
    fleet = lipmpc.UnknownEnvFleet(rings, N_horizon=3, lidar_range=1.5)
    runs = []
    for seed in (3, 3, 4):
        r = fleet.run(st0, goal, foot, K, noise_seed=seed)
        torch.cuda.synchronize()
        runs.append({k: v.cpu().numpy().copy() for k, v in r.items()})
    assert np.array_equal(runs[0]["X_pred"], runs[1]["X_pred"]) and np.array_equal(runs[0]["n_steps"], runs[1]["n_steps"])
    assert not np.array_equal(runs[0]["X_pred"], runs[2]["X_pred"])
    assert runs[0]["n_steps"].min() >= 5


@pytest.mark.gpu
def test_gpu_per_robot_maps_and_input_limits(golden_dir):
    """env_shared = 0 (one true map per robot, include/lipmpc.h) gives each robot exactly what the shared-map call
    gives it on its own map; more than 384 obstacles in range, or a ring longer than v_env, is flagged as overflow and
    never read past the kernel's candidate list; bad arguments are refused before any launch."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    import lipmpc
    d = np.load(os.path.join(golden_dir, "lidar_golden.npz"))
    lr = float(d["lidar_range"][0])
    cases = [i for i in range(len(d["pos"])) if float(d["lidar_range"][i]) == lr][:6]
    B = len(cases)
    assert B >= 3
    maps, poss, noises = [], [], []
    for i in cases:
        pos, rings, rng = _case(d, i)
        maps.append(rings); poss.append(pos); noises.append(d["noise"][i])
    n_env = max(len(m) for m in maps); v_env = max(len(r) for m in maps for r in m)
    exy = np.zeros((B, n_env, v_env, 2)); env = np.zeros((B, n_env), np.int32)
    for i, m in enumerate(maps):
        for j, r in enumerate(m):
            exy[i, j, :len(r)] = r; env[i, j] = len(r)
    st = np.zeros((B, 5)); st[:, 0] = [p[0] for p in poss]; st[:, 2] = [p[1] for p in poss]
    sensor = lipmpc.LidarSensor(maps[0], lidar_range=lr, n_obs_max=12, v_max=40)
    d_st, d_noise = torch.as_tensor(st, device="cuda"), torch.as_tensor(np.array(noises), device="cuda")
    out = sensor.sense(d_st, d_noise, with_debug=True, env_xy=torch.as_tensor(exy, device="cuda"), env_nv=torch.as_tensor(env, device="cuda"))
    torch.cuda.synchronize()
    for i in range(B):
        own = lipmpc.LidarSensor(maps[i], lidar_range=lr, n_obs_max=12, v_max=40)
        o = own.sense(d_st[i:i + 1].contiguous(), d_noise[i:i + 1].contiguous(), with_debug=True)
        torch.cuda.synchronize()
        for k in ("n_inferred", "overflow", "obs_nv", "labels"):
            assert torch.equal(out[k][i], o[k][0]), (i, k)
        n = int(o["n_inferred"][0])
        for j in range(n):
            nv = int(o["obs_nv"][0, j])
            assert torch.equal(out["obs_xy"][i, j, :nv], o["obs_xy"][0, j, :nv])
        assert n == int((d["inf_nv"][cases[i]] > 0).sum())        # and it is the reference's answer for that map
    with pytest.raises(ValueError):
        sensor.sense(d_st, d_noise, env_xy=torch.as_tensor(exy[:2], device="cuda"), env_nv=torch.as_tensor(env[:2], device="cuda"))
    # 500 small squares around the robot, all in range: the candidate list holds 384 -> overflow flag, no fault
    rng = np.random.default_rng(1)
    ang = rng.uniform(0, 2 * np.pi, 500); rad = rng.uniform(0.5, 1.4, 500)
    sq = np.array([[-0.01, -0.01], [0.01, -0.01], [0.01, 0.01], [-0.01, 0.01]])
    many = [np.array([rad[k] * np.cos(ang[k]), rad[k] * np.sin(ang[k])]) + sq for k in range(500)]
    crowded = lipmpc.LidarSensor(many, lidar_range=1.5, n_obs_max=12, v_max=32)
    o = crowded.sense(torch.zeros((3, 5), dtype=torch.float64, device="cuda"), None)
    torch.cuda.synchronize()
    assert o["overflow"].cpu().tolist() == [1, 1, 1]
    # the same map with 300 obstacles fits the list (whatever the clustering says about slots)
    fits = lipmpc.LidarSensor(many[:300], lidar_range=1.5, n_obs_max=50, v_max=32)
    o = fits.sense(torch.zeros((1, 5), dtype=torch.float64, device="cuda"), None, with_debug=True)
    torch.cuda.synchronize()
    assert int((~torch.isnan(o["hits"][0, :, 0])).sum()) > 100
    # a ring count beyond v_env is clamped and flagged
    bad_nv = torch.as_tensor(np.full((1, 1), 9, np.int32), device="cuda")
    one = lipmpc.LidarSensor([sq + 1.0], lidar_range=3.0)
    o = one.sense(torch.zeros((1, 5), dtype=torch.float64, device="cuda"), None,
                  env_xy=torch.as_tensor((sq + 1.0)[None, None], device="cuda").contiguous(), env_nv=bad_nv)
    torch.cuda.synchronize()
    assert int(o["overflow"][0]) == 1
    # argument errors never reach the device
    lib = lipmpc._lib.load()
    z = C.c_void_p(0)
    assert lib.lipmpc_lidar_sense_batch(0, 1, 400, 0, 1, 1, C.c_double(1.5), C.c_double(0.3), 3, 12, 32, z, z, z, z, z, z, z, z, z, z, z, z) == -1
    assert lib.lipmpc_lidar_sense_batch(0, 1, 360, 0, 1, 1, C.c_double(1.5), C.c_double(0.3), 3, 12, 99, z, z, z, z, z, z, z, z, z, z, z, z) == -1


@pytest.mark.gpu
def test_gpu_more_clusters_than_slots(golden_dir):
    """A ring of 14 separate posts around the robot = 14 clusters for 12 default slots: the drop-in class scans again into
    the large layout and plans against all 14 (the reference constrains against every inferred obstacle,
    HumanoidMPCUnknownEnvironment.py:55-64); the fleet loop stops such a robot with STATUS_SENSOR_OVERFLOW instead of
    walking it through obstacles it sensed."""
    torch = pytest.importorskip("torch")
    import lipmpc
    posts = []
    for k in range(14):                       # 0.60 m between centres, 0.48 m between surfaces: 14 DBSCAN clusters (eps 0.3)
        a = 2 * np.pi * k / 14
        c = np.array([1.35 * np.cos(a), 1.35 * np.sin(a)])
        posts.append(c + 0.06 * np.array([[np.cos(t), np.sin(t)] for t in np.linspace(0, 2 * np.pi, 9)[:-1]]))
    mpc = lipmpc.HumanoidMPCUnknownEnvironment(goal=(5, 0.3), obstacles=posts, N_horizon=3, N_mpc_timesteps=3, sampling_time=0.4,
                                               init_state=(0, 0, 0, 0, 0), verbosity=0, lidar_range=1.5, noise_seed=2)
    X, U, _ = mpc.run_simulation(None, make_fast_plot=False, fill_animator=False)
    assert len(mpc.list_inferred_obstacles[0]) == 14 and X.shape[1] >= 2
    fleet = lipmpc.UnknownEnvFleet(posts, N_horizon=3, lidar_range=1.5, n_obs_max=12)
    st0 = torch.zeros((2, 5), dtype=torch.float64, device="cuda"); st0[1, 0] = 4.0        # robot 1 sees only a few posts
    goal = torch.tensor([[5.0, 0.3]] * 2, dtype=torch.float64, device="cuda")
    r = fleet.run(st0, goal, torch.ones((2,), dtype=torch.int8, device="cuda"), 4, noise_seed=2)
    torch.cuda.synchronize()
    assert int(r["n_steps"][0]) == 0 and int(r["last_status"][0]) == lipmpc.STATUS_SENSOR_OVERFLOW and int(r["overflow"][0]) == 1
    assert int(r["n_steps"][1]) == 4 and int(r["overflow"][1]) == 0


@pytest.mark.gpu
def test_gpu_sensor_overflow_reaches_the_step_status(golden_dir):
    """A scan whose clusters do not fit the obstacle slots must not hand out a usable-looking plan (the reference constrains
    against every inferred obstacle, HumanoidMPCUnknownEnvironment.py:55-64): through the one-call step
    (lipmpc_sense_plan_step_batch) and through sense() + plan_step_batch_c_eta(overflow=) the robot's status is
    SENSOR_OVERFLOW, its outputs NaN, and lipmpc_advance_batch -- which looks at status only -- leaves it where it is;
    a robot of the same batch whose scan fits is planned and advanced as usual."""
    torch = pytest.importorskip("torch")
    import lipmpc
    posts = []
    for k in range(6):                        # six separate posts around the origin: six clusters for ONE slot
        a = 2 * np.pi * k / 6
        c = np.array([1.0 * np.cos(a), 1.0 * np.sin(a)])
        posts.append(c + 0.06 * np.array([[np.cos(t), np.sin(t)] for t in np.linspace(0, 2 * np.pi, 9)[:-1]]))
    sensor = lipmpc.LidarSensor(posts, lidar_range=1.5, n_obs_max=1, v_max=32)
    sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=3, n_obs_max=1, v_max=32))
    st = torch.zeros((2, 5), dtype=torch.float64, device="cuda"); st[1, 0] = 2.3          # robot 1 sees one post only
    goal = torch.tensor([[5.0, 0.3]] * 2, dtype=torch.float64, device="cuda")
    foot = torch.ones((2,), dtype=torch.int8, device="cuda")
    noise = 0.01 * torch.randn((2, 360, 2), dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    sen, out = sensor.sense_plan_step(sv, st, goal, foot, noise)
    torch.cuda.synchronize()
    assert sen["overflow"].tolist() == [1, 0] and int(sen["n_inferred"][1]) == 1
    assert out["status"].tolist() == [lipmpc.STATUS_SENSOR_OVERFLOW, lipmpc.STATUS_SOLVED]
    assert bool(torch.isnan(out["U"][0]).all()) and bool(torch.isnan(out["X"][0]).all()) and bool(torch.isfinite(out["U"][1]).all())
    st2, foot2 = st.clone(), foot.clone()
    sv.advance(st2, foot2, out)
    torch.cuda.synchronize()
    assert torch.equal(st2[0], st[0]) and int(foot2[0]) == 1                 # not walked against obstacles it sensed but dropped
    assert not torch.equal(st2[1], st[1]) and int(foot2[1]) == -1
    # the two-call form: the flags travel with the rows
    sen2 = sensor.sense(st, noise, c_eta=True, rings=False)
    o2 = sv.plan_step_batch_c_eta(st, goal, foot, sen2["c_eta"], overflow=sen2["overflow"])
    torch.cuda.synchronize()
    assert o2["status"].tolist() == out["status"].tolist() and torch.equal(o2["U"][1], out["U"][1])
    # without the flags the truncated list is solved as given (the caller's responsibility, documented)
    o3 = sv.plan_step_batch_c_eta(st, goal, foot, sen2["c_eta"])
    torch.cuda.synchronize()
    assert int(o3["status"][0]) in (lipmpc.STATUS_SOLVED, lipmpc.STATUS_UNCERTIFIED)


@pytest.mark.gpu
def test_gpu_constraint_assembly_fused_into_the_scan(golden_dir):
    """lipmpc_lidar_c_eta_batch (scan -> clusters -> hulls -> closest point / normal in ONE launch, hulls in LDS) against
    (a) the oracle chain lidar oracle -> closest_point_and_normal on the reference's golden scans and (b), for 4096
    robots, the two-launch path: the (c, eta) rows are bit-identical to what the step kernel's front end derives from
    the rings lipmpc_lidar_sense_batch writes, and so is everything the step solves from them."""
    torch = pytest.importorskip("torch")
    import lipmpc
    import lipmpc_oracle as O
    d = np.load(os.path.join(golden_dir, "lidar_golden.npz"))
    tab = L.ray_table()
    worst = 0.0
    for i in range(0, len(d["pos"]), 2):
        pos, rings, rng = _case(d, i)
        sensor = lipmpc.LidarSensor(rings, lidar_range=rng, resolution=360, n_obs_max=12, v_max=40)
        st = torch.tensor([[pos[0], 0.0, pos[1], 0.0, 0.0]], dtype=torch.float64, device="cuda")
        out = sensor.sense(st, torch.as_tensor(d["noise"][i][None], device="cuda"), c_eta=True)
        torch.cuda.synchronize()
        n = int(out["n_inferred"][0]); ce = out["c_eta"][0].cpu().numpy()
        assert n == int((d["inf_nv"][i] > 0).sum()) and int(out["overflow"][0]) == 0
        assert np.all(ce[n:] == 0.0)
        for j in range(n):
            ref_ring = d["inf_xy"][i][j][: d["inf_nv"][i][j]]                       # Qhull's ring, from the reference
            c, eta, _, degen = O.closest_point_and_normal(np.array([pos[0], pos[1]]), ref_ring)
            assert not degen
            worst = max(worst, float(np.max(np.abs(ce[j, :2] - c))), float(np.max(np.abs(ce[j, 2:] - eta))))
    assert worst < 1e-13, worst
    # 4096 robots on one map: fused constraint assembly == rings -> front end of the step kernel, bit for bit
    _, rings, _ = _case(d, 0)
    B, N = 4096, 3
    rng = np.random.default_rng(11)
    pos = rng.uniform(-0.8, 5.8, (B, 2))
    st = np.zeros((B, 5)); st[:, 0] = pos[:, 0]; st[:, 2] = pos[:, 1]; st[:, 4] = rng.uniform(-1, 1, B)
    d_st = torch.as_tensor(st, device="cuda")
    noise = torch.as_tensor(0.01 * rng.standard_normal((B, 360, 2)), device="cuda")
    sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, n_obs_max=12, v_max=32)
    two = sensor.sense(d_st, noise)
    one = sensor.sense(d_st, noise, c_eta=True)
    lean = sensor.sense(d_st, noise, c_eta=True, rings=False)                       # hulls never leave the kernel
    sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=12, v_max=32))
    goal = torch.tensor([[5.0, 5.0]], dtype=torch.float64, device="cuda").repeat(B, 1).contiguous()
    foot = torch.ones((B,), dtype=torch.int8, device="cuda")
    delta = torch.as_tensor(rng.choice([0.0, 0.05], B), device="cuda")
    ref = sv.plan_step_batch(d_st, goal, foot, two["obs_xy"], two["obs_nv"], delta, with_c_eta=True)
    got = sv.plan_step_batch_c_eta(d_st, goal, foot, lean["c_eta"], delta)
    torch.cuda.synchronize()
    assert "obs_xy" not in lean
    for k in ("n_inferred", "overflow", "obs_nv"):
        assert torch.equal(one[k], two[k]), k
    assert torch.equal(one["c_eta"], lean["c_eta"]) and torch.equal(one["n_inferred"], lean["n_inferred"])
    has = (two["obs_nv"] > 0)
    assert torch.equal(one["obs_xy"][has], two["obs_xy"][has])
    assert torch.equal(torch.nan_to_num(one["c_eta"][has], nan=7.0), torch.nan_to_num(ref["c_eta"][has], nan=7.0))
    assert int(has.sum()) > 2 * B and not bool(torch.any(one["c_eta"][~has] != 0.0))
    st_ref, st_got = ref["status"].cpu().numpy(), got["status"].cpu().numpy()
    assert np.array_equal(st_ref, st_got)
    for k in ("iters", "active", "theta", "omega"):
        assert torch.equal(ref[k], got[k]), k
    ok = torch.as_tensor((st_ref == 0) | (st_ref == 4), device="cuda")
    for k in ("U", "X", "obj"):
        assert torch.equal(ref[k][ok], got[k][ok]), k
    assert int(ok.sum()) > 0.6 * B
    # the launch order is a pure scheduling matter: with an order buffer the call ranks its robots itself (estimated reading
    # counts, heaviest first) before it scans them -- same answers as in index order, launch after launch
    sched = sensor.make_schedule(B)
    for rep in range(3):
        got_s = sensor.sense(d_st, noise, c_eta=True, rings=False, schedule=sched)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(got_s["c_eta"], nan=7.0), torch.nan_to_num(lean["c_eta"], nan=7.0)), rep
        assert torch.equal(got_s["n_inferred"], lean["n_inferred"]) and torch.equal(got_s["overflow"], lean["overflow"])
        sc = sched.cpu().numpy()
        assert sc[0] == B and np.array_equal(np.sort(sc[2:2 + B]), np.arange(B))               # a complete order of the B robots ...
        # ... by descending estimated reading count, dealt out boustrophedon over rounds of one position per SIMD (positions one
        # round apart share a SIMD while the whole grid is resident: each gets a heavy robot with a light one)
        period = 4 * torch.cuda.get_device_properties(0).multi_processor_count
        ranked = sc[2:2 + B].copy()
        for q in range(1, B // period, 2):
            ranked[q * period:(q + 1) * period] = ranked[q * period:(q + 1) * period][::-1]
        assert np.all(np.diff(sc[2 + B:][ranked]) <= 0)
    with pytest.raises(ValueError):
        sensor.sense(d_st[:100].contiguous(), noise[:100].contiguous(), c_eta=True, schedule=sched)
    # the whole step of the unknown-environment variant in one C call
    sen1, out1 = sensor.sense_plan_step(sv, d_st, goal, foot, noise, delta, schedule=sched)
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(sen1["c_eta"], nan=7.0), torch.nan_to_num(lean["c_eta"], nan=7.0))
    for k in ("status", "iters", "active"):
        assert torch.equal(out1[k], got[k]), k
    assert torch.equal(out1["U"][ok], got["U"][ok])
    # a NaN normal in the rows (degenerate geometry met by the producer) is reported as DEGENERATE, like the ring front end
    ce = lean["c_eta"][:4].clone(); ce[1, 0, 2] = float("nan")
    r = sv.plan_step_batch_c_eta(d_st[:4].contiguous(), goal[:4].contiguous(), foot[:4].contiguous(), ce.contiguous())
    torch.cuda.synchronize()
    assert int(r["status"][1]) == lipmpc.STATUS_DEGENERATE


@pytest.mark.gpu
@pytest.mark.parametrize("lidar_range,resolution,seed", [(1.0, 360, 0), (1.5, 360, 1), (3.0, 360, 2), (1.5, 180, 3), (2.0, 90, 4)])
def test_gpu_lidar_fuzz_against_oracle(lidar_range, resolution, seed):
    """Random maps, ranges and resolutions, robots anywhere (also inside obstacles): readings bit-identical to the oracle
    (which is pinned to the reference's range_finder), DBSCAN labels equal, hull rings equal, (c, eta) rows within 1e-12 of
    the oracle's closest point on its own ring."""
    torch = pytest.importorskip("torch")
    import lipmpc
    import lipmpc_oracle as O
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    rng = np.random.default_rng(seed)
    n_env = int(rng.integers(6, 21))
    exy, env = synth.synthetic_fields(1, n_env, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=100 + seed, delta=0.6)
    rings = [exy[0, j, : env[0, j]] for j in range(n_env) if env[0, j] > 0]
    B = 24
    pos = rng.uniform(-1.0, 6.0, (B, 2))
    st = np.zeros((B, 5)); st[:, 0] = pos[:, 0]; st[:, 2] = pos[:, 1]
    noise = 0.01 * rng.standard_normal((B, resolution, 2))
    sensor = lipmpc.LidarSensor(rings, lidar_range=lidar_range, resolution=resolution, n_obs_max=16, v_max=48)
    out = sensor.sense(torch.as_tensor(st, device="cuda"), torch.as_tensor(noise, device="cuda"), with_debug=True, c_eta=True)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in out.items()}
    tab = L.ray_table(resolution)
    n_rings = 0
    for b in range(B):
        hits, valid, labels, inferred = L.range_finder(pos[b], rings, lidar_range, noise=noise[b], table=tab)
        gv = ~np.isnan(g["hits"][b, :, 0])
        assert np.array_equal(gv, valid), b
        assert np.array_equal(g["hits"][b][valid], hits[valid]), b                      # bit-exact readings
        assert np.array_equal(g["labels"][b][valid], labels), b                         # cluster labels
        assert np.all(g["labels"][b][~valid] == -2)
        if g["overflow"][b]:
            continue
        assert g["n_inferred"][b] == len(inferred), (b, g["n_inferred"][b], len(inferred))
        for j, ring in enumerate(inferred):
            assert _same_ring(g["obs_xy"][b, j, : g["obs_nv"][b, j]], ring), (b, j)
            c, eta, _, degen = O.closest_point_and_normal(pos[b], ring)
            if not degen:
                assert np.max(np.abs(g["c_eta"][b, j, :2] - c)) < 1e-12 and np.max(np.abs(g["c_eta"][b, j, 2:] - eta)) < 1e-12
            n_rings += 1
    assert n_rings > B // 2


@pytest.mark.gpu
@pytest.mark.parametrize("eps,min_samples,field", [(0.3, 3, "crowded"), (0.3, 2, "crowded"), (0.3, 4, "crowded"), (0.3, 5, "crowded"),
                                                   (0.3, 1, "crowded"), (0.3, 7, "crowded"), (0.12, 3, "crowded"), (0.6, 3, "crowded"),
                                                   (0.04, 3, "crowded"), (0.02, 2, "crowded"), (0.3, 3, "pickets"), (0.12, 3, "pickets"),
                                                   (0.3, 5, "pickets"), (0.3, 3, "far"), (0.3, 4, "far")])
def test_gpu_clustering_routes_against_oracle(monkeypatch, eps, min_samples, field):
    """The scan clusters by chains of consecutive readings where it can prove that this is DBSCAN's answer, and by neighbour
    rows where it cannot (csrc/lipmpc_lidar_chains.inc / _rows.inc): 512 robots anywhere on a crowded map (inside obstacles too)
    or in a field of 90 pickets, over eps / min_samples that make the proof succeed for nearly all of them, for some, and for
    none -- the labels of every reading equal
    oracle/lidar_oracle.py::dbscan_labels on the scan's own readings, and the rings are the oracle's hulls of those clusters."""
    torch = pytest.importorskip("torch")
    import lipmpc
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    lidar_mod = import_module("humanoid-navigation-using-mpc-ldcbf_amd.lidar")
    monkeypatch.setattr(lidar_mod, "DBSCAN_EPS", eps)
    monkeypatch.setattr(lidar_mod, "DBSCAN_MIN_SAMPLES", min_samples)
    rng = np.random.default_rng(int(eps * 1000) + min_samples)
    lidar_range, resolution = (3.0, 180) if field == "far" else (1.5, 360)      # "far": sparse readings on far walls, gaps near eps
    if field != "pickets":
        exy, env = synth.synthetic_fields(1, 20, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=9, delta=0.6)
        rings = [exy[0, j, : env[0, j]] for j in range(20) if env[0, j] > 0]
    else:
        # a field of pickets: 90 squares of 0.12 m, 0.35-0.9 m apart -- scans of many short pieces (more than the chain route's
        # eight, pieces shorter than min_samples next to longer ones, gaps around eps): the general route and its hand-over
        cs = rng.uniform(-1.0, 6.0, (90, 2))
        sq = 0.06 * np.array([[-1.0, -1.0], [1.0, -1.0], [1.0, 1.0], [-1.0, 1.0]])
        rings = [c + sq for c in cs]
    B = 512
    pos = rng.uniform(-1.0, 6.0, (B, 2))
    st = np.zeros((B, 5)); st[:, 0] = pos[:, 0]; st[:, 2] = pos[:, 1]
    noise = 0.01 * rng.standard_normal((B, resolution, 2))
    sensor = lipmpc.LidarSensor(rings, lidar_range=lidar_range, resolution=resolution, n_obs_max=24, v_max=64)
    out = sensor.sense(torch.as_tensor(st, device="cuda"), torch.as_tensor(noise, device="cuda"), with_debug=True, c_eta=True)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in out.items()}
    n_clusters = n_rings = 0
    for b in range(B):
        valid = ~np.isnan(g["hits"][b, :, 0])
        pts = g["hits"][b][valid]
        if len(pts) == 0:
            assert g["n_inferred"][b] == 0
            continue
        labels = L.dbscan_labels(pts, eps, min_samples)
        assert np.array_equal(g["labels"][b][valid], labels), (b, len(pts))
        n_clusters += labels.max() + 1
        if g["overflow"][b]:
            continue
        want = [r for r in (L.hull_ring(pts[labels == k]) for k in range(labels.max() + 1)) if r is not None]
        assert g["n_inferred"][b] == len(want), (b, g["n_inferred"][b], len(want))
        for j, ring in enumerate(want):
            assert _same_ring(g["obs_xy"][b, j, : g["obs_nv"][b, j]], ring), (b, j)
            n_rings += 1
    assert n_clusters > 0 and (n_rings > B // 4 or eps < 0.1 or field == "pickets")


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 63, 1500, 5000, 9000])
def test_gpu_launch_order_any_batch_size(B):
    """The call's own ranking (lidar_weight_kernel -> lidar_order_kernel: ranks dealt out boustrophedon over rounds of one launch
    position per SIMD, the last, partial round as ranked) is a complete order of the robots at every batch size -- below one
    round, between rounds, beyond what is resident at once -- and leaves every answer as it is in index order."""
    torch = pytest.importorskip("torch")
    import lipmpc
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    exy, env = synth.synthetic_fields(1, 20, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=9, delta=0.6)
    rings = [exy[0, j, : env[0, j]] for j in range(20) if env[0, j] > 0]
    sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, resolution=360, n_obs_max=12, v_max=32)
    gen = torch.Generator(device="cuda").manual_seed(B)
    pos = torch.rand((B, 2), dtype=torch.float64, device="cuda", generator=gen) * 7.0 - 1.0
    st = torch.zeros((B, 5), dtype=torch.float64, device="cuda"); st[:, 0] = pos[:, 0]; st[:, 2] = pos[:, 1]
    noise = 0.01 * torch.randn((B, 360, 2), dtype=torch.float64, device="cuda", generator=gen)
    plain = sensor.sense(st, noise, c_eta=True, rings=False, schedule=None)
    sched = sensor.make_schedule(B)
    sched.fill_(-7)                                                       # scratch: whatever it holds
    ranked = sensor.sense(st, noise, c_eta=True, rings=False, schedule=sched)
    torch.cuda.synchronize()
    for k in ("n_inferred", "overflow"):
        assert torch.equal(plain[k], ranked[k]), k
    assert torch.equal(torch.nan_to_num(plain["c_eta"], nan=7.0), torch.nan_to_num(ranked["c_eta"], nan=7.0))
    sc = sched.cpu().numpy()
    assert sc[0] == B and np.array_equal(np.sort(sc[2:2 + B]), np.arange(B))
