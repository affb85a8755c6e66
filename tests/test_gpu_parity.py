"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on
identical inputs.  Tolerances: footsteps/CoM trajectories 1e-5 (north_star; observed ~1e-9),
theta/omega 1e-12, c/eta 4e-15 (eta: /distance), status and active-set indices bit-exact
(problems whose interior-point margin |log(z/s)| is below 0.5 are excluded from the bit-exact
active-set comparison and counted)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import lipmpc  # noqa: E402
import lipmpc_oracle as O  # noqa: E402
from helpers import (IPOPT_LIKE_TOL, TIGHT_BAND, assert_active_sets, closed_loop_problems, compare_active_sets,  # noqa: E402
                     load_rings)


def _dev(a, dt):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")


def run_gpu(problems, N, n_obs_max, v_max, flags=0, with_c_eta=True, sampling_time=0.4):
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs_max, v_max=v_max, flags=flags, sampling_time=sampling_time)
    sv = lipmpc.BatchedLipMpc(P)
    st = np.array([p[0] for p in problems])
    goal = np.array([p[1] for p in problems], float)
    foot = np.array([p[2] for p in problems], np.int8)
    xy, nv = lipmpc.pack_rings([p[3] for p in problems], n_obs_max, v_max)
    delta = np.array([p[4] for p in problems], float)
    out = sv.plan_step_batch(_dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8),
                             _dev(xy, torch.float64) if n_obs_max else None,
                             _dev(nv, torch.int32) if n_obs_max else None, _dev(delta, torch.float64),
                             with_c_eta=with_c_eta and n_obs_max > 0, with_diag=True, with_working=True)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res["active_bits"] = lipmpc.unpack_active(res["active"], P.num_rows)
    res["working_bits"] = lipmpc.unpack_active(res["working"], P.num_rows)
    return res


def compare(problems, res, N, exact=True, tol_u=1e-5):
    P = O.Params(N=N)
    worst_u = worst_x = 0.0
    n_weak = n_act_cmp = n_work_diff = 0
    it_diff = 0
    for b, (st, goal, s0, obs, delta) in enumerate(problems):
        r = O.plan_step(st, goal, s0, obs, delta, P, exact=exact)
        assert res["status"][b] == r["status"], (b, res["status"][b], r["status"])
        assert np.max(np.abs(res["theta"][b] - r["theta"])) < 1e-12
        assert np.max(np.abs(res["omega"][b] - r["omega"])) < 1e-12
        if r["status"] not in (O.STATUS_SOLVED, O.STATUS_UNCERTIFIED):
            assert np.all(np.isnan(res["U"][b]))
            continue
        du = np.max(np.abs(res["U"][b] - r["U"]))
        dx = np.max(np.abs(res["X"][b] - r["X"]))
        worst_u, worst_x = max(worst_u, du), max(worst_x, dx)
        assert du < tol_u and dx < tol_u, (b, du, dx)
        assert abs(res["obj"][b] - r["obj"]) < 1e-6 * max(1.0, abs(r["obj"]))
        it_diff = max(it_diff, abs(int(res["iters"][b]) - r["iters"]))
        if exact and r["status"] == O.STATUS_SOLVED:
            # `active` = the primal tight set of the optimum (slack <= 1e-7: unique), bit for bit -- unless a row sits within
            # TIGHT_BAND x the distance between the two answers of that tolerance on either side (helpers.compare_active_sets)
            band = TIGHT_BAND * max(du, dx)
            if r["tight_margin"] < band or res["diag"][b][4] < band:
                n_weak += 1
            else:
                n_act_cmp += 1
                assert np.array_equal(res["active_bits"][b], r["active"]), (b, np.where(res["active_bits"][b] != r["active"]))
            # the finish's working sets: equal wherever both certificates are decisive at 1e-7; a difference below that needs
            # the two answers within 1e-6
            if not np.array_equal(res["working_bits"][b], r["working"]):
                n_work_diff += 1
                assert min(r.get("cert_margin", 0.0), res["diag"][b][3]) < 1e-7 and du < 1e-6, (b, r.get("cert_margin"), res["diag"][b][3], du)
    return dict(worst_u=worst_u, worst_x=worst_x, n_weak=n_weak, n_act_cmp=n_act_cmp, it_diff=it_diff, n_work_diff=n_work_diff)


def test_smoke_single_problem_no_obstacles():
    probs = [(np.array([0.0, 0, 0.0, 0, 0.0]), (5.0, 5.0), 1, [], 0.0)]
    res = run_gpu(probs, 3, 0, 5)
    s = compare(probs, res, 3)
    assert s["worst_u"] < 1e-8


@pytest.mark.parametrize("N,n_obs,ntraj,steps", [(3, 3, 6, 25), (5, 3, 4, 20), (8, 10, 6, 25), (8, 13, 4, 20), (10, 12, 5, 14)])
def test_closed_loop_states_match_oracle(N, n_obs, ntraj, steps):
    probs = list(closed_loop_problems(N, n_obs, ntraj, steps, seed=100 + N))
    assert len(probs) > 50
    res = run_gpu(probs, N, n_obs, 5)
    s = compare(probs, res, N)
    print("parity", N, n_obs, len(probs), s)
    assert s["worst_u"] < 1e-7 and s["it_diff"] <= (1 if N <= 8 else 2)     # two DPP rows per problem: other summation order
    assert s["n_act_cmp"] >= 0.97 * (s["n_act_cmp"] + s["n_weak"]) and s["n_act_cmp"] > 0.8 * len(probs)


def test_reference_generator_fields_config2(golden_dir):
    """BASELINE config 2 inputs: obstacle fields produced by the reference's own generate_obstacles
    (fixture fields_cfg2.npz), N=8, 10 obstacles, delta in {0, 0.3}."""
    d = np.load(os.path.join(golden_dir, "fields_cfg2.npz"))
    fields = [[d["rings"][f][j][: d["nv"][f][j]] for j in range(10)] for f in range(24)]
    probs = []
    for delta, sl in [(0.0, slice(0, 12)), (0.3, slice(12, 24))]:
        probs += list(closed_loop_problems(8, 10, 12, 16, seed=7, delta=delta, fields=fields[sl]))
    res = run_gpu(probs, 8, 10, 5)
    s = compare(probs, res, 8)
    print("cfg2 parity", len(probs), s)
    assert s["worst_u"] < 1e-7


def test_interior_flag_matches_oracle_ipm():
    """FLAG_INTERIOR returns the interior-point iterate (no finish).  That point is only defined up
    to the stop tolerance: |q - q*| <= sqrt(m mu) ~ 4e-4 in weakly determined directions, and a
    one-iteration difference between two correct implementations moves it by that much; so the
    bound here is the iterate's own accuracy, not the 1e-5 of the exact path."""
    probs = list(closed_loop_problems(8, 10, 3, 20, seed=3))
    res = run_gpu(probs, 8, 10, 5, flags=lipmpc.FLAG_INTERIOR)
    s = compare(probs, res, 8, exact=False, tol_u=4e-4)
    # strictly interior: every predicted CoM keeps a positive LDCBF value
    P = O.Params(N=8)
    for b, (st, goal, s0, obs, delta) in enumerate(probs):
        r = O.plan_step(st, goal, s0, obs, delta, P, exact=False)
        p = res["X"][b][1:, [0, 2]]
        for c, eta in zip(r["c"], r["eta"]):
            assert np.all((p - c) @ eta - delta > 0.0)


def test_geometry_golden_through_c_eta(golden_dir):
    """c, eta of the kernel's front end against the REFERENCE's own outputs (geometry_golden.npz)."""
    d = np.load(os.path.join(golden_dir, "geometry_golden.npz"))
    probs = []
    for q, w in zip(d["pts"], d["which"]):
        ring = d["rings"][w][: d["nv"][w]]
        probs.append((np.array([q[0], 0.0, q[1], 0.0, 0.0]), (q[0] + 5.0, q[1] + 5.0), 1, [ring], 0.0))
    res = run_gpu(probs, 3, 1, 24)
    ce = res["c_eta"][:, 0, :]
    assert np.max(np.abs(ce[:, :2] - d["c"])) <= 4e-15
    dist = np.hypot(d["pts"][:, 0] - d["c"][:, 0], d["pts"][:, 1] - d["c"][:, 1])
    assert np.all(np.max(np.abs(ce[:, 2:] - d["eta"]), axis=1) <= 2e-15 + 4e-15 / dist)
    # inside flag = sign of eta.(x - c): reference flips eta for inside points -> the k=0 row is negative
    h0 = np.sum(ce[:, 2:] * (d["pts"] - ce[:, :2]), axis=1)
    assert np.array_equal(h0 < 0, d["inside"])
    # and bit-for-bit against the oracle's ring-order restatement
    for b in range(0, len(probs), 7):
        c2, e2, _, _ = O.closest_point_and_normal(d["pts"][b], probs[b][3][0])
        assert np.array_equal(ce[b, :2], c2) and np.array_equal(ce[b, 2:], e2)


def test_status_codes_and_empty_slots():
    sq = np.array([[1.0, 1.0], [2.0, 1.0], [2.0, 2.0], [1.0, 2.0]])
    far = np.array([[7.0, 7.0], [8.0, 7.0], [7.5, 8.0]])
    probs = [
        (np.array([1.5, 0, 1.4, 0, 0.0]), (5.0, 5.0), 1, [sq, far], 0.0),     # inside -> infeasible
        (np.array([1.0, 0, 1.0, 0, 0.0]), (5.0, 5.0), 1, [sq, far], 0.0),     # on a vertex -> degenerate
        (np.array([0.0, 0, 0.0, 0, 0.0]), (5.0, 5.0), -1, [far], 0.0),        # one empty slot
        (np.array([0.0, 0, 0.0, 0, 0.0]), (5.0, 5.0), 1, [np.array([[1.0, 1], [1, 1], [2, 2]]), far], 0.0),
    ]
    res = run_gpu(probs, 3, 2, 5)
    assert list(res["status"]) == [2, 3, 0, 3]
    P = O.Params(N=3)
    r = O.plan_step(probs[2][0], probs[2][1], -1, [far], 0.0, P)
    assert np.max(np.abs(res["U"][2] - r["U"])) < 1e-8
    # canonical indices of the present obstacle's rows follow n_obs_max = 2 slots
    act = res["active_bits"][2]
    ora = np.zeros(P.N * 9 + 4 * 2, bool)
    ora[: 9 * 3] = r["active"][: 9 * 3]
    for k in range(4):
        ora[27 + 2 * k] = r["active"][27 + k]
    assert np.array_equal(act, ora)


def test_api_errors():
    with pytest.raises(RuntimeError):
        lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=3, finish_rounds=-1))
    import ctypes as C
    lib = lipmpc._lib.load()
    p = lipmpc.LipMpcParams(N=40).to_c()
    h = C.c_void_p()
    assert lib.lipmpc_create(C.byref(p), 0, C.byref(h)) == -2
    assert b"unsupported" in lib.lipmpc_strerror(-2)
    sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=3))
    with pytest.raises(ValueError):
        sv.plan_step_batch(torch.zeros((2, 5)), torch.zeros((2, 2)), torch.zeros(2, dtype=torch.int8))


def test_compat_class_closed_loop(golden_dir):
    """HumanoidMPC drop-in on the reference's circles scenario (simulation_1.py:85-102): same first
    steps as the PDF-recovered reference run, same run length as the oracle loop."""
    obs = load_rings(os.path.join(golden_dir, "scenario_circles.npz"))
    mpc = lipmpc.HumanoidMPC(goal=(6, -3), obstacles=obs, N_horizon=3, N_mpc_timesteps=300, sampling_time=0.4,
                             init_state=(0, 0, 3, 0, 0), verbosity=0)
    X, U, anim = mpc.run_simulation(path_to_gif=None, make_fast_plot=False, plot_animation=False, fill_animator=False)
    assert anim is None and X.shape[0] == 5 and U.shape[0] == 3 and X.shape[1] == U.shape[1] + 1
    pdf = np.load(os.path.join(golden_dir, "pdf_series.npz"))
    ex = pdf["Simulation1Circles/ev0/s0"][:, 1] + 6.0
    assert abs(X.shape[1] - len(ex)) <= 3
    assert np.max(np.abs(X[0, :3] - ex[:3])) < 5e-7
    Xo, Uo = O.run_closed_loop((6, -3), obs, N_horizon=3, N_mpc_timesteps=300, sampling_time=0.4,
                               init_state=(0, 0, 3, 0, 0), exact=False, params=O.Params(tol_interior=IPOPT_LIKE_TOL))
    n = min(10, Xo.shape[1], X.shape[1])
    assert np.max(np.abs(X[:, :n] - Xo[:, :n])) < 1e-5


def test_subgoal_sequencing_matches_oracle_handoff(golden_dir):
    """HumanoidMPCWithRRT hand-off (HumanoidMPCWithRRT.py:155-181) with given sub-goals: a fresh closed loop per
    sub-goal from the previous run's last state, outputs concatenated; against the oracle loop chained the same
    way, and the batched rollout_subgoals against the class robot by robot (different numbers of sub-goals)."""
    obs = load_rings(os.path.join(golden_dir, "scenario_circles.npz"))
    subs = np.array([[1.0, -2.0], [4.0, -2.5], [6.0, -3.0]])
    kw = dict(N_horizon=3, N_mpc_timesteps=40, sampling_time=0.4)
    mpc = lipmpc.HumanoidMPCWithRRT(goal=(6, -3), obstacles=obs, init_state=(9, 9, 9, 9, 9), verbosity=0,
                                    sub_goals=subs, **kw)
    X, U, _ = mpc.run_simulation(None, make_fast_plot=False, fill_animator=False)
    Xo, Uo, st = None, None, (0, 0, 0, 0, 0)           # the reference ignores init_state here (:155)
    lens = []
    for g in subs:
        xs, us = O.run_closed_loop(tuple(g), obs, init_state=st, exact=False, params=O.Params(tol_interior=IPOPT_LIKE_TOL), **kw)
        st = tuple(xs[:, -1])
        lens.append(xs.shape[1])
        Xo = xs if Xo is None else np.concatenate((Xo, xs), axis=1)
        Uo = us if Uo is None else np.concatenate((Uo, us), axis=1)
    assert X.shape == Xo.shape and U.shape == Uo.shape and len(lens) == 3 and min(lens) > 3
    assert np.max(np.abs(X - Xo)) < 1e-5 and np.max(np.abs(U - Uo)) < 1e-5
    assert np.hypot(X[0, -1] - 6.0, X[2, -1] + 3.0) < 0.2          # arrived
    with pytest.raises(ImportError):
        lipmpc.HumanoidMPCWithRRT(goal=(6, -3), obstacles=obs, verbosity=0, **kw).run_simulation(None)
    # batched: robot 0 walks all three sub-goals, robot 1 two, robot 2 one
    P = lipmpc.LipMpcParams(N=3, n_obs_max=len(obs), v_max=max(len(r) for r in obs),
                            flags=lipmpc.FLAG_INTERIOR, tol_interior=IPOPT_LIKE_TOL)
    sv = lipmpc.BatchedLipMpc(P)
    xy, nv = lipmpc.pack_rings([obs] * 3, P.n_obs_max, P.v_max)
    n_sub = torch.tensor([3, 2, 1], dtype=torch.int32)
    out = sv.rollout_subgoals(_dev(np.zeros((3, 5)), torch.float64), _dev(np.repeat(subs[None], 3, 0), torch.float64),
                              n_sub, _dev(np.ones(3, np.int8), torch.int8), _dev(xy, torch.float64),
                              _dev(nv, torch.int32), None, k_max=40, mpc_step=1)
    torch.cuda.synchronize()
    nk = out["n_kept"].cpu().numpy()
    Xb = out["X_pred"].cpu().numpy()
    assert nk[1, 2] == -1 and nk[2, 1] == -1 and nk[2, 2] == -1
    for b, ns in enumerate([3, 2, 1]):
        cat = np.concatenate([Xb[b, s, : nk[b, s] + 1].T for s in range(ns)], axis=1)
        ref = Xo[:, : sum(lens[:ns])]
        assert cat.shape == ref.shape and np.max(np.abs(cat - ref)) < 1e-5


def test_full_size_batch_against_c_oracle():
    """BASELINE config 2 at full size: B=4096, N=8, 10 obstacles (bench.py's generator and on-device
    walk), every problem compared with the dense C oracle; plus size-independent properties:
    idempotence of the launch, feasibility of every returned trajectory, dynamics consistency."""
    import c_oracle
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    B, N, n_obs = 4096, 8, 10
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5)
    sv = lipmpc.BatchedLipMpc(P)
    walker = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=lipmpc.FLAG_INTERIOR))
    xy, nv = synth.synthetic_fields(B, n_obs, 0.5, 9.5, (0.0, 0.0), (10.0, 10.0), seed=4242)
    obs_xy, obs_nv = _dev(xy, torch.float64), _dev(nv, torch.int32)
    goal = torch.tensor([[10.0, 10.0]], dtype=torch.float64, device="cuda").repeat(B, 1).contiguous()
    delta = torch.zeros((B,), dtype=torch.float64, device="cuda")
    state, foot = synth.walk_states(walker, obs_xy, obs_nv, goal, 30, seed=7, delta=delta)
    out = sv.plan_step_batch(state, goal, foot, obs_xy, obs_nv, delta, with_diag=True, with_working=True)
    out2 = sv.plan_step_batch(state, goal, foot, obs_xy, obs_nv, delta)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in out.items()}
    assert np.array_equal(g["U"], out2["U"].cpu().numpy(), equal_nan=True)          # deterministic / idempotent
    ref = c_oracle.plan_step_batch(P, state.cpu().numpy(), goal.cpu().numpy(), foot.cpu().numpy(), xy, nv,
                                   delta.cpu().numpy(), n_threads=8)
    same = g["status"] == ref["status"]
    assert same.mean() > 0.995, (np.sum(~same), np.unique(g["status"][~same]), np.unique(ref["status"][~same]))
    ok = same & (ref["status"] == 0)
    assert ok.mean() > 0.97
    assert np.max(np.abs(g["U"][ok] - ref["U"][ok])) < 1e-5        # north_star tolerance (observed ~1e-8)
    assert np.max(np.abs(g["X"][ok] - ref["X"][ok])) < 1e-5
    assert np.max(np.abs(g["theta"] - ref["theta"])) < 1e-12
    info, _ = compare_active_sets(ok, g, ref)
    print("full-size batch:", info)
    assert_active_sets("full-size batch", info, 0.99)               # active-constraint indices bit-exact
    # properties that need no oracle: LIP dynamics hold along every returned trajectory ...
    A_, B_ = O.lip_matrices(O.Params(N=N))
    X, U = g["X"][ok], g["U"][ok]
    for k in range(N):
        assert np.max(np.abs(X[:, k + 1] - (X[:, k] @ A_.T + U[:, k] @ B_.T))) < 1e-9
    # ... and every predicted CoM respects every LDCBF half-space (c, eta from the same launch)
    ce = sv.plan_step_batch(state, goal, foot, obs_xy, obs_nv, delta, with_c_eta=True)["c_eta"].cpu().numpy()[ok]
    p = X[:, 1:, :][:, :, [0, 2]]
    hval = np.einsum("bkc,bjc->bkj", p, ce[:, :, 2:]) - np.sum(ce[:, :, 2:] * ce[:, :, :2], axis=2)[:, None, :]
    assert hval.min() > -1e-8


def test_two_row_groups_horizon12_finish_rounds_match_oracle():
    """N=12 (G=32: a problem spans two DPP rows, exchanges cross rows by v_permlane16_swap) with 10 obstacles:
    statuses must equal the C oracle's problem by problem, interior-point iteration counts and active-set finish
    rounds almost always -- the add/drop sequence of the finish depends on every cross-row argmin."""
    import c_oracle
    N, n_obs = 12, 10
    probs = list(closed_loop_problems(N, n_obs, 8, 25, seed=5))
    res = run_gpu(probs, N, n_obs, 5)
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5)
    xy, nv = lipmpc.pack_rings([p[3] for p in probs], n_obs, 5)
    ref = c_oracle.plan_step_batch(P, np.array([p[0] for p in probs]), np.array([p[1] for p in probs], float),
                                   np.array([p[2] for p in probs], np.int8), xy, nv,
                                   np.array([p[4] for p in probs], float), n_threads=8)
    assert np.array_equal(res["status"], ref["status"])
    # the two-row factorisation uses the symmetric (Cholesky-form) update, the oracle the LDL^T form: equal up to
    # rounding, which in the ill-conditioned last iterations (cond K ~ 1e15) moves a few counts by one
    di = np.abs(res["iters"] - ref["iters"])
    assert di.max() <= 1 and (di == 0).mean() >= 0.9, (di.max(), (di == 0).mean())
    assert (res["diag"][:, 0] == ref["diag"][:, 0]).mean() >= 0.97         # finish rounds
    assert (ref["diag"][:, 0] >= 3).sum() >= 10                            # the multi-round path is exercised
    ok = ref["status"] == 0
    assert np.max(np.abs(res["U"][ok] - ref["U"][ok])) < 1e-5
    assert np.max(np.abs(res["X"][ok] - ref["X"][ok])) < 1e-5


def test_finish_rounds_cap_matches_oracle():
    """lipmpc_params.finish_rounds bounds the add/drop rounds of the finish (tail-latency control): with any cap
    the statuses equal the C oracle's under the same cap, no problem reports more rounds than the cap, and a tighter
    cap only ever turns SOLVED into UNCERTIFIED (whose answer is the interior-point iterate, still within 1e-5)."""
    import c_oracle
    N, n_obs = 8, 10
    probs = list(closed_loop_problems(N, n_obs, 8, 30, seed=3))
    st = np.array([p[0] for p in probs]); goal = np.array([p[1] for p in probs], float)
    foot = np.array([p[2] for p in probs], np.int8); delta = np.array([p[4] for p in probs], float)
    xy, nv = lipmpc.pack_rings([p[3] for p in probs], n_obs, 5)
    res = {}
    for cap in (10, 0, 2, 1):
        P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, finish_rounds=cap)
        sv = lipmpc.BatchedLipMpc(P)
        out = sv.plan_step_batch(_dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8),
                                 _dev(xy, torch.float64), _dev(nv, torch.int32), _dev(delta, torch.float64), with_diag=True)
        torch.cuda.synchronize()
        g = {k: v.cpu().numpy() for k, v in out.items()}
        ref = c_oracle.plan_step_batch(P, st, goal, foot, xy, nv, delta, n_threads=8)
        assert np.array_equal(g["status"], ref["status"]), cap
        assert g["diag"][:, 0].max() <= (cap if cap else 5)
        res[cap] = g
    s10, s5, s2, s1 = (res[c]["status"] for c in (10, 0, 2, 1))
    for a, b in ((s10, s5), (s5, s2), (s2, s1)):
        assert np.all((a == b) | ((a == 0) & (b == 4)))
    assert (s1 == 4).sum() > (s10 == 4).sum()
    both = (s10 == 0) & (s1 == 4)
    assert np.max(np.abs(res[10]["U"][both] - res[1]["U"][both])) < 1e-5


def test_limit_cycle_case_on_gpu(golden_dir):
    """The step that sends plain Mehrotra iterations into a 2-cycle (tests/golden/limit_cycle_case.npz,
    test_oracle.py::test_limit_cycle_case_converges): solved by the kernel in the oracle's iteration count."""
    d = np.load(os.path.join(golden_dir, "limit_cycle_case.npz"))
    prob = (d["state"], tuple(d["goal"]), 1, [d["ring0"]], 0.0)
    res = run_gpu([prob] * 5, 3, 12, 32)
    r = O.plan_step(d["state"], d["goal"], 1, [d["ring0"]], 0.0, O.Params(N=3), exact=True)
    assert np.all(res["status"] == 0) and r["status"] == 0
    assert np.all(res["iters"] == r["iters"]) and r["iters"] <= 20
    assert np.max(np.abs(res["U"][0] - r["U"])) < 1e-8


@pytest.mark.parametrize("N,n_obs", [(N, n) for N in (6, 12) for n in (0, 3, 9, 14, 22, 40)] + [(3, n) for n in (0, 3, 9, 14)])
def test_every_instantiation_against_c_oracle(N, n_obs):
    """One small batch through each of the sixteen kernel instantiations (16 / 32 lanes per problem x 0, 2, 5, 7
    register row slots and 13, 25 streamed ones, + the four half-size factorisations horizons up to 4 run on), step kernel
    and rollout kernel: statuses as the C oracle's,
    footsteps within 1e-5; the rollout's first sample equals the step kernel's answer.  (The compiler has
    miscompiled single instantiations after unrelated source changes — every one of them is pinned here.)"""
    import c_oracle
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    B = 96
    rng = np.random.default_rng(100 * N + n_obs)
    if n_obs:
        xy, nv = synth.synthetic_fields(B, n_obs, 0.5, 12.0, (0.0, 0.0), (12.5, 12.5), seed=7 + n_obs)
    else:
        xy, nv = np.zeros((B, 0, 5, 2)), np.zeros((B, 0), np.int32)
    st = np.zeros((B, 5)); st[:, 0] = rng.uniform(0, 1.5, B); st[:, 2] = rng.uniform(0, 1.5, B)
    st[:, 1] = rng.uniform(0.0, 0.3, B); st[:, 3] = np.where(rng.random(B) < 0.5, 0.2, -0.2); st[:, 4] = rng.uniform(0.3, 1.2, B)
    foot = np.where(st[:, 3] > 0, 1, -1).astype(np.int8)
    goal = np.tile([[12.5, 12.5]], (B, 1))
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5)
    sv = lipmpc.BatchedLipMpc(P)
    d = lambda a, dt: _dev(a, dt) if a.size else None
    out = sv.plan_step_batch(_dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8),
                             d(xy, torch.float64), d(nv, torch.int32), None)
    ro = sv.rollout(_dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8),
                    d(xy, torch.float64), d(nv, torch.int32), None, k_max=2, mpc_step=1)
    torch.cuda.synchronize()
    ref = c_oracle.plan_step_batch(P, st, goal, foot, xy if n_obs else None, nv if n_obs else None, None, n_threads=8)
    gs, U = out["status"].cpu().numpy(), out["U"].cpu().numpy()
    # solved-or-not must agree exactly; certified (0) vs uncertified (4) may differ on a few ill-conditioned problems
    rs = ref["status"]
    assert np.array_equal(np.isin(gs, (0, 4)), np.isin(rs, (0, 4))), (np.bincount(gs, minlength=5), np.bincount(rs, minlength=5))
    assert np.array_equal(gs[~np.isin(gs, (0, 4))], rs[~np.isin(rs, (0, 4))])
    assert (gs == rs).mean() >= 0.97
    ok = (gs == 0) & (rs == 0)
    assert ok.sum() >= B // 2
    assert np.max(np.abs(U[ok] - ref["U"][ok])) < 1e-5
    ur = ro["U_pred"].cpu().numpy()[:, 0, :2]
    solved = (gs == 0) | (gs == 4)
    assert np.array_equal(ro["n_steps"].cpu().numpy() >= 1, solved)
    assert np.max(np.abs(ur[ok] - U[ok, 0])) < 1e-7


def _rollout_inputs(n_robots, n_obs, seed):
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    xy, nv = synth.synthetic_fields(n_robots, n_obs, 0.5, 9.5, (0.0, 0.0), (10.0, 10.0), seed=seed)
    st = np.zeros((n_robots, 5))
    goal = np.tile([[10.0, 10.0]], (n_robots, 1))
    foot = np.ones(n_robots, np.int8)
    return st, goal, foot, xy, nv


@pytest.mark.parametrize("N,n_obs", [(8, 10), (8, 12), (12, 12), (8, 20)])
def test_rollout_equals_host_driven_loop(N, n_obs):
    """lipmpc_rollout_batch (whole closed loop in one launch) against the same loop driven from the host
    with plan_step_batch + advance_batch: same step code, so the first samples agree to rounding and
    the runs have the same structure (the loop is chaotic afterwards: LDCBF normals amplify 1e-16).
    The parameter sets walk through the rollout instantiations: 5 and 7 register row slots per lane at 16 lanes,
    7 at 32 lanes, 13 streamed rows."""
    B, K = 64, 40
    st, goal, foot, xy, nv = _rollout_inputs(B, n_obs, 11)
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=lipmpc.FLAG_INTERIOR)
    sv = lipmpc.BatchedLipMpc(P)
    d_st, d_goal, d_foot = _dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8)
    d_xy, d_nv = _dev(xy, torch.float64), _dev(nv, torch.int32)
    ro = sv.rollout(d_st, d_goal, d_foot, d_xy, d_nv, None, k_max=K, mpc_step=1)
    torch.cuda.synchronize()
    Xr, Ur, nr = ro["X_pred"].cpu().numpy(), ro["U_pred"].cpu().numpy(), ro["n_steps"].cpu().numpy()
    # host-driven
    s, f = d_st.clone(), d_foot.clone()
    Xh = np.zeros((B, K + 1, 5)); Uh = np.zeros((B, K, 3)); nh = np.zeros(B, int)
    Xh[:, 0] = st
    alive = np.ones(B, bool); last_obj = np.full(B, np.inf)
    out = sv.alloc_outputs(B)
    for k in range(K):
        alive &= ~(last_obj < 0.05)
        sv.plan_step_batch(s, d_goal, f, d_xy, d_nv, None, out=out)
        status = out["status"].cpu().numpy()
        ok = (status == 0) | (status == 4)
        alive &= ok
        last_obj = np.where(alive, out["obj"].cpu().numpy(), last_obj)
        Uh[:, k, :2] = out["U"][:, 0].cpu().numpy(); Uh[:, k, 2] = out["omega"][:, 0].cpu().numpy()
        sv.advance(s, f, out)
        Xh[:, k + 1] = s.cpu().numpy()
        nh += alive
    assert np.array_equal(nr, nh) or np.mean(np.abs(nr - nh) <= 2) > 0.9
    # 16-lane groups with register rows: both kernels compile the step identically.  32-lane groups and streamed
    # rows: the two kernels schedule (and contract) the step differently, the interior iterate moves in its last
    # digits and the loop amplifies that
    tol_x, tol_u, n_cmp = (1e-9, 1e-7, 8) if (N <= 8 and n_obs <= 14) else (1e-5, 1e-5, 5)
    for b in range(B):
        n = min(nr[b], nh[b], n_cmp)
        assert np.max(np.abs(Xr[b, : n + 1] - Xh[b, : n + 1])) < tol_x, b
        assert np.max(np.abs(Ur[b, :n] - Uh[b, :n])) < tol_u, b
    assert np.median(nr) >= 30          # robots actually walk


def test_rollout_against_oracle_closed_loop(golden_dir):
    """Device rollout vs the numpy oracle's run_closed_loop on the reference's circles scenario
    (simulation_1.py:85-102), delta = 0 and 0.3, and with sampling_time = 0.1 (mpc_step = 4)."""
    obs = load_rings(os.path.join(golden_dir, "scenario_circles.npz"))
    for delta, samp, kmax in [(0.0, 0.4, 300), (0.3, 0.4, 300), (0.0, 0.1, 160)]:
        mpc = lipmpc.HumanoidMPCCustomLCBF(goal=(6, -3), obstacles=obs, N_horizon=3, N_mpc_timesteps=kmax // max(1, int(0.4 / samp)),
                                           sampling_time=samp, init_state=(0, 0, 3, 0, 0), verbosity=0,
                                           distance_from_obstacles=delta)
        X, U, _ = mpc.run_simulation(None, make_fast_plot=False, fill_animator=False)
        Xo, Uo = O.run_closed_loop((6, -3), obs, N_horizon=3, N_mpc_timesteps=kmax // max(1, int(0.4 / samp)),
                                   sampling_time=samp, init_state=(0, 0, 3, 0, 0), delta=delta, exact=False,
                                   params=O.Params(tol_interior=IPOPT_LIKE_TOL))
        assert X.shape[0] == 5 and U.shape[0] == 3 and X.shape[1] == U.shape[1] + 1
        n = min(12 * max(1, int(0.4 / samp)), X.shape[1], Xo.shape[1])
        assert np.max(np.abs(X[:, :n] - Xo[:, :n])) < 1e-6, (delta, samp)
        assert np.max(np.abs(U[:, : n - 1] - Uo[:, : n - 1])) < 1e-5
        if samp == 0.4:
            assert abs(X.shape[1] - Xo.shape[1]) <= 3
            assert np.hypot(X[0, -1] - 6, X[2, -1] + 3) < 0.3       # reached the goal region


def test_per_problem_bounds_overrides_match_oracle():
    """bounds[B,4] = (V_MAX_x, V_MAX_y, ALPHA, OMEGA_MAX) per problem — the knobs bounds_tuning.py:17-26
    mutates in `conf` — against the C oracle given the same overrides."""
    import c_oracle
    rng = np.random.default_rng(5)
    probs = list(closed_loop_problems(3, 3, 6, 20, seed=77))
    B = len(probs)
    bounds = np.stack([rng.uniform(0.5, 1.0, B), rng.uniform(0.25, 0.4, B), rng.uniform(0.5, 4.0, B),
                       rng.uniform(0.4, 1.0, B)], axis=1)
    P = lipmpc.LipMpcParams(N=3, n_obs_max=3, v_max=5)
    sv = lipmpc.BatchedLipMpc(P)
    st = np.array([p[0] for p in probs]); goal = np.array([p[1] for p in probs], float)
    foot = np.array([p[2] for p in probs], np.int8)
    xy, nv = lipmpc.pack_rings([p[3] for p in probs], 3, 5)
    out = sv.plan_step_batch(_dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8),
                             _dev(xy, torch.float64), _dev(nv, torch.int32), None, bounds=_dev(bounds, torch.float64))
    torch.cuda.synchronize()
    ref = c_oracle.plan_step_batch(P, st, goal, foot, xy, nv, None, bounds=bounds)
    status = out["status"].cpu().numpy()
    assert np.array_equal(status, ref["status"])
    ok = status == 0
    assert ok.sum() > 0.5 * B
    assert np.max(np.abs(out["U"].cpu().numpy()[ok] - ref["U"][ok])) < 1e-7
    assert np.max(np.abs(out["omega"].cpu().numpy() - ref["omega"])) < 1e-12
    # the override really changes the problem
    base = sv.plan_step_batch(_dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8),
                              _dev(xy, torch.float64), _dev(nv, torch.int32), None)
    assert np.nanmax(np.abs(base["U"].cpu().numpy() - out["U"].cpu().numpy())) > 1e-3


def test_bounds_tuning_sweep_in_one_launch():
    """The reference's hyper-parameter sweep (report_simulations/bounds_tuning.py:13-45): 16 x 4 x 35 x 12 =
    26,880 closed-loop simulations (N=3, sampling_time 0.1 -> mpc_step 4, 300 MPC steps, goal (5,5), no
    obstacles), scored by mean |v_y| over the first 50 samples among runs ending within 1 m of the goal.
    Here: ONE rollout launch, checked robot-by-robot against the oracle's closed loop on a sample of the grid.
    (The reference records its own winner in a comment, bounds_tuning.py:72: (0.85, 0.2, 2.3, 0.8).  With exact
    solves that robot's 7th MPC step is infeasible by 4e-3 -- the lateral-velocity band [0.1, 0.2] is left after
    the three rotation-only samples -- so the recorded winner cannot be used as a golden; it is printed.)"""
    import itertools
    vx = np.arange(0.2, 1, 0.05); vy = np.arange(0.2, 0.4, 0.05); al = np.arange(0.5, 4, 0.1); om = np.arange(0.4, 1, 0.05)
    combos = np.array(list(itertools.product(vx, vy, al, om)))
    default = np.array([[0.8, 0.4, 3.6, 0.156 * np.pi]])
    combos = np.vstack([combos, default])
    B = len(combos)
    assert B == 26881
    P = lipmpc.LipMpcParams(N=3, n_obs_max=0, v_max=5, sampling_time=0.1, flags=lipmpc.FLAG_INTERIOR)
    sv = lipmpc.BatchedLipMpc(P)
    st = torch.zeros((B, 5), dtype=torch.float64, device="cuda")
    goal = torch.tensor([[5.0, 5.0]], dtype=torch.float64, device="cuda").repeat(B, 1).contiguous()
    foot = torch.ones((B,), dtype=torch.int8, device="cuda")
    ro = sv.rollout(st, goal, foot, None, None, None, k_max=1200, mpc_step=4, bounds=_dev(combos, torch.float64))
    torch.cuda.synchronize()
    n = ro["n_steps"].long()
    X = ro["X_pred"]
    last = torch.where(n < 1200, n, torch.full_like(n, 1199))     # the reference's X[:, :k+1] truncation
    idx = torch.arange(B, device="cuda")
    endp = X[idx, last][:, [0, 2]]
    reached = ((endp - 5.0) ** 2 <= 1.0).all(dim=1)
    cols = torch.arange(50, device="cuda")[None, :]
    valid = cols <= last[:, None]
    score = (X[:, :50, 3].abs() * valid).sum(1) / valid.sum(1)
    score = torch.where(reached, score, torch.full_like(score, float("inf")))
    score_np, reached_np, n_np = score.cpu().numpy(), reached.cpu().numpy(), n.cpu().numpy()
    best = int(np.argmin(score_np))
    ref_i = int(np.argmin(np.abs(combos - np.array([0.85, 0.2, 2.3, 0.8])).sum(1)))
    print("bounds_tuning: reached", int(reached_np.sum()), "of", B, "| best", combos[best], score_np[best],
          "| reference's recorded winner", combos[ref_i], "samples", n_np[ref_i], "reached", bool(reached_np[ref_i]))
    assert reached_np[-1] and 250 < n_np[-1] < 320            # the default configuration walks to the goal
    assert reached_np.sum() > 500
    assert combos[best][1] == combos[reached_np][:, 1].min()   # the score favours the smallest lateral bound that still arrives
    # robot-by-robot against the oracle's closed loop (first 3 MPC steps tight; outcome for early deaths)
    rng = np.random.default_rng(0)
    Xh = X.cpu().numpy()
    for b in list(rng.choice(B - 1, 12, replace=False)) + [B - 1, ref_i]:
        Pn = O.Params(N=3, sampling_time=0.1, v_max=(combos[b][0], combos[b][1]), alpha=combos[b][2], omega_max=combos[b][3])
        Xo, Uo = O.run_closed_loop((5, 5), [], N_horizon=3, N_mpc_timesteps=300, sampling_time=0.1,
                                   init_state=(0, 0, 0, 0, 0), params=Pn, exact=False)
        m = min(12, Xo.shape[1] - 1, n_np[b])
        assert np.max(np.abs(Xh[b, : m + 1].T - Xo[:, : m + 1])) < 1e-6, b
        if Xo.shape[1] - 1 < 100:                                # a robot that dies early dies at the same sample
            assert abs(n_np[b] - (Xo.shape[1] - 1)) <= 4, (b, n_np[b], Xo.shape[1] - 1)


def test_edge_shapes_and_ragged_inputs(golden_dir):
    """Batch sizes that do not fill a wavefront (1, 2, 3, 5 problems; 4 or 2 problems share a wave), horizons
    1 and 2, 24-vertex rings (v_max = 32) next to triangles with empty obstacle slots in between, B = 0."""
    import c_oracle
    circles = load_rings(os.path.join(golden_dir, "scenario_circles.npz"))      # 9-, 19- and 24-gons
    tri = np.array([[7.0, 7.0], [8.0, 7.0], [7.5, 8.0]])
    for N in (1, 2, 3):
        for B in (1, 2, 3, 5):
            probs = []
            for b in range(B):
                obs = [circles[b % 3], tri] if b % 2 == 0 else [tri]            # second slot empty on odd problems
                probs.append((np.array([0.1 * b, 0.0, 3.0 - 0.2 * b, 0.0, -0.1 * b]), (6.0, -3.0), 1 if b % 2 == 0 else -1, obs, 0.05 * b))
            res = run_gpu(probs, N, 2, 32)
            P = lipmpc.LipMpcParams(N=N, n_obs_max=2, v_max=32)
            xy, nv = lipmpc.pack_rings([p[3] for p in probs], 2, 32)
            ref = c_oracle.plan_step_batch(P, np.array([p[0] for p in probs]), np.array([p[1] for p in probs], float),
                                           np.array([p[2] for p in probs], np.int8), xy, nv,
                                           np.array([p[4] for p in probs], float))
            assert np.array_equal(res["status"], ref["status"]), (N, B)
            ok = ref["status"] == 0
            assert ok.all()
            assert np.max(np.abs(res["U"] - ref["U"])) < 1e-8 and np.max(np.abs(res["X"] - ref["X"])) < 1e-8
            assert np.array_equal(res["c_eta"], ref["c_eta"])
            assert np.array_equal(lipmpc.unpack_active(res["active"], P.num_rows), lipmpc.unpack_active(ref["active"], P.num_rows))
    # B = 0 is a no-op, not an error
    sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=3))
    out = sv.plan_step_batch(torch.zeros((0, 5), dtype=torch.float64, device="cuda"),
                             torch.zeros((0, 2), dtype=torch.float64, device="cuda"),
                             torch.zeros((0,), dtype=torch.int8, device="cuda"))
    assert out["U"].shape == (0, 3, 2)


def test_given_half_spaces_entry_point_and_subclass_hooks(golden_dir):
    """lipmpc_plan_step_batch_c_eta (the reference's _get_list_c_and_eta / _compute_single_lcbf hooks as data,
    HumanoidMpc.py:252-261, 296-319): (1) fed with the c / eta the ring entry point reports, it returns the same bits;
    (2) with arbitrary (non-unit) normals and empty slots it matches the C oracle given the same half-spaces;
    (3) a subclass that subtracts a margin in _compute_single_lcbf the way HumanoidMPCCustomLCBF.py:30-31 does walks
    the same trajectory as the class with distance_from_obstacles, and one that drops an obstacle in
    _get_list_c_and_eta the trajectory of the scenario without that obstacle."""
    import c_oracle
    N, n_obs = 8, 10
    probs = list(closed_loop_problems(N, n_obs, 4, 14, seed=21, delta=0.1))
    B = len(probs)
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5)
    sv = lipmpc.BatchedLipMpc(P)
    st = _dev(np.array([p[0] for p in probs]), torch.float64); goal = _dev(np.array([p[1] for p in probs], float), torch.float64)
    foot = _dev(np.array([p[2] for p in probs], np.int8), torch.int8); delta = _dev(np.array([p[4] for p in probs], float), torch.float64)
    xy, nv = lipmpc.pack_rings([p[3] for p in probs], n_obs, 5)
    a = sv.plan_step_batch(st, goal, foot, _dev(xy, torch.float64), _dev(nv, torch.int32), delta, with_c_eta=True)
    b = sv.plan_step_batch_c_eta(st, goal, foot, a["c_eta"].contiguous(), delta)
    torch.cuda.synchronize()
    for k in ("U", "X", "obj", "status", "iters", "active"):
        assert torch.equal(a[k], b[k]) or (k in ("U", "X", "obj") and np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy(), equal_nan=True)), k
    rng = np.random.default_rng(3)
    ce = a["c_eta"].cpu().numpy().copy()
    ce[:, :, 2:] *= rng.uniform(0.5, 2.0, (B, n_obs, 1))          # non-unit normals
    ce[:, 7:, 2:] = 0.0                                            # three empty slots
    g = sv.plan_step_batch_c_eta(st, goal, foot, _dev(ce, torch.float64), delta)
    torch.cuda.synchronize()
    ref = c_oracle.plan_step_batch(P, st.cpu().numpy(), goal.cpu().numpy(), foot.cpu().numpy(), None, None, delta.cpu().numpy(),
                                   c_eta_in=ce, n_threads=8)
    gs = g["status"].cpu().numpy()
    assert np.array_equal(gs, ref["status"])
    ok = gs == 0
    assert ok.sum() > 0.8 * B and np.max(np.abs(g["U"].cpu().numpy()[ok] - ref["U"][ok])) < 1e-7
    act = lipmpc.unpack_active(g["active"].cpu().numpy(), P.num_rows)
    assert not act[:, 9 * N:].reshape(B, N + 1, n_obs)[:, :, 7:].any()       # rows of empty slots are never active
    # (3) subclass hooks
    obs = load_rings(os.path.join(golden_dir, "scenario_circles.npz"))
    kw = dict(goal=(6, -3), N_horizon=3, N_mpc_timesteps=25, sampling_time=0.4, init_state=(0, 0, 3, 0, 0), verbosity=0)

    class Margin(lipmpc.HumanoidMPC):
        def _compute_single_lcbf(self, x, eta, c):
            return super()._compute_single_lcbf(x, eta, c) - 0.3

    class Blind(lipmpc.HumanoidMPC):
        def _get_list_c_and_eta(self, x_k, y_k):
            cs, etas = super()._get_list_c_and_eta(x_k, y_k)
            return cs[:2], etas[:2]

    Xm, Um, _ = Margin(obstacles=obs, **kw).run_simulation(None, make_fast_plot=False, fill_animator=False)
    Xc, Uc, _ = lipmpc.HumanoidMPCCustomLCBF(obstacles=obs, distance_from_obstacles=0.3, **kw).run_simulation(None, make_fast_plot=False, fill_animator=False)
    assert Xm.shape == Xc.shape and np.max(np.abs(Xm - Xc)) < 1e-7 and np.max(np.abs(Um - Uc)) < 1e-6
    X0, U0, _ = lipmpc.HumanoidMPC(obstacles=obs, **kw).run_simulation(None, make_fast_plot=False, fill_animator=False)
    assert np.max(np.abs(Xm - X0)) > 1e-3                                   # the margin really changes the walk
    Xb, Ub, _ = Blind(obstacles=obs, **kw).run_simulation(None, make_fast_plot=False, fill_animator=False)
    X2, U2, _ = lipmpc.HumanoidMPC(obstacles=obs[:2], **kw).run_simulation(None, make_fast_plot=False, fill_animator=False)
    assert Xb.shape == X2.shape and np.max(np.abs(Xb - X2)) < 1e-7


def test_rollout_warm_start_matches_oracle_and_saves_iterations(golden_dir):
    """LIPMPC_FLAG_WARM_START: every step of the on-device loop starts from the previous step's interior-point result
    shifted by one stage (oracle: shift_warm_start / run_closed_loop(warm_start=True); the reference seeds its next solve
    with the shifted prediction, HumanoidMpc.py:450-455).  Same trajectory as the oracle's warm-started loop, fewer
    iterations than the cold loop on the same robots."""
    obs = load_rings(os.path.join(golden_dir, "scenario_circles.npz"))
    kw = dict(N_horizon=3, N_mpc_timesteps=300, sampling_time=0.4, init_state=(0, 0, 3, 0, 0))
    mpc = lipmpc.HumanoidMPC(goal=(6, -3), obstacles=obs, verbosity=0, warm_start=True, **kw)
    X, U, _ = mpc.run_simulation(None, make_fast_plot=False, fill_animator=False)
    Xo, Uo = O.run_closed_loop((6, -3), obs, exact=False, params=O.Params(tol_interior=IPOPT_LIKE_TOL), warm_start=True, **kw)
    it_warm_oracle = float(np.mean(O.run_closed_loop.last_iters))
    n = min(12, X.shape[1], Xo.shape[1])
    assert abs(X.shape[1] - Xo.shape[1]) <= 3 and np.max(np.abs(X[:, :n] - Xo[:, :n])) < 1e-6
    assert np.hypot(X[0, -1] - 6, X[2, -1] + 3) < 0.3
    # batch: 256 robots on random fields, N = 8, cold against warm, same robots
    st, goal, foot, xy, nv = _rollout_inputs(256, 10, 5)
    res = {}
    for name, fl in (("cold", lipmpc.FLAG_INTERIOR), ("warm", lipmpc.FLAG_INTERIOR | lipmpc.FLAG_WARM_START)):
        sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=8, n_obs_max=10, v_max=5, flags=fl))
        ro = sv.rollout(_dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8), _dev(xy, torch.float64),
                        _dev(nv, torch.int32), None, k_max=40, mpc_step=1)
        torch.cuda.synchronize()
        res[name] = (int(ro["n_steps"].sum()), int(ro["total_iters"].sum()), ro["X_pred"].cpu().numpy(), ro["n_steps"].cpu().numpy())
    per = {k: v[1] / v[0] for k, v in res.items()}
    print("rollout iterations per step: cold %.2f, warm %.2f (oracle, circles, warm: %.2f)" % (per["cold"], per["warm"], it_warm_oracle))
    assert per["warm"] < 0.95 * per["cold"]
    assert abs(res["warm"][0] - res["cold"][0]) < 0.05 * res["cold"][0]          # the robots walk as far
    both = (res["cold"][3] >= 5) & (res["warm"][3] >= 5)
    assert np.max(np.abs(res["cold"][2][both, :3] - res["warm"][2][both, :3])) < 1e-4     # same optimum: first steps agree to the stop tolerance


@pytest.mark.parametrize("N,n_obs", [(1, 0), (1, 5), (2, 9), (3, 3), (4, 14), (5, 6), (8, 10), (8, 13), (12, 10), (12, 13), (16, 30), (16, 50)])
def test_fuzz_odd_inputs_against_c_oracle(N, n_obs):
    """Arbitrary, mostly odd inputs (empty obstacle slots, zero-length edges, robots inside obstacles, random velocities,
    headings and goals) through every kind of instantiation (horizons 1..16, the half-size factorisation, register and
    streamed rows): statuses equal the C oracle's problem by problem, footsteps of the solved ones within 1e-6.
    (tests/dev/fuzz_gpu.py is the long version: 170 k problems, profiles/r02_dev_tools/r02_fuzz.txt.)"""
    import c_oracle
    from importlib import import_module
    synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
    rng = np.random.default_rng(1000 * N + n_obs)
    B = 384
    xy, nv = synth.synthetic_fields(64, max(n_obs, 1), 0.5, 9.5, (0, 0), (10, 10), seed=int(rng.integers(1e6)))
    xy, nv = xy[:, :n_obs], nv[:, :n_obs]
    idx = rng.integers(0, 64, B)
    xy, nv = xy[idx].copy(), nv[idx].copy()
    nv[rng.random(nv.shape) < 0.1] = 0                                  # empty slots
    if n_obs:
        deg = rng.random(B) < 0.02
        xy[deg, 0, 1] = xy[deg, 0, 0]                                   # zero-length edges
    st = np.zeros((B, 5)); st[:, 0] = rng.uniform(0, 10, B); st[:, 2] = rng.uniform(0, 10, B)
    st[:, 1] = rng.normal(0, 0.3, B); st[:, 3] = rng.normal(0, 0.3, B); st[:, 4] = rng.uniform(-4, 4, B)
    calm = rng.random(B) < 0.5
    st[calm, 1] *= 0.1; st[calm, 3] = np.where(rng.random(calm.sum()) < 0.5, 0.25, -0.25)
    goal = rng.uniform(-2, 12, (B, 2)); foot = rng.choice([-1, 1], B).astype(np.int8)
    delta = np.where(rng.random(B) < 0.5, 0.0, rng.uniform(0, 0.5, B))
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5)
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    out = lipmpc.BatchedLipMpc(P).plan_step_batch(dev(st, torch.float64), dev(goal, torch.float64), dev(foot, torch.int8),
                                                  dev(xy, torch.float64) if n_obs else None, dev(nv, torch.int32) if n_obs else None,
                                                  dev(delta, torch.float64))
    torch.cuda.synchronize()
    ref = c_oracle.plan_step_batch(P, st, goal, foot, xy if n_obs else None, nv if n_obs else None, delta, n_threads=8)
    gs = out["status"].cpu().numpy()
    assert np.array_equal(gs, ref["status"]), (N, n_obs, np.bincount(gs, minlength=5), np.bincount(ref["status"], minlength=5))
    ok = gs == 0
    assert ok.sum() > B // 2
    U = out["U"].cpu().numpy()
    assert not np.isnan(U[ok]).any() and np.max(np.abs(U[ok] - ref["U"][ok])) < 1e-6
    assert np.isnan(U[np.isin(gs, (1, 2, 3))]).all()                    # unsolved problems carry NaN, never stale numbers


def test_presolve_gating_is_one_rule():
    """The presolve runs unless a flag says otherwise -- LIPMPC_FLAG_INTERIOR, LIPMPC_FLAG_WARM_START (the interior iterates
    matter there) or LIPMPC_FLAG_NO_PRESOLVE -- and that ONE rule holds in the kernel's front end, in the launcher's choice of
    kernel and in both oracles, whether or not the step at hand has a warm start to read.  A plain plan_step on a handle with
    LIPMPC_FLAG_WARM_START therefore keeps every row: bit-identical to the LIPMPC_FLAG_NO_PRESOLVE handle, iteration counts
    those of the C oracle under the same flag (they differ from the presolved path's)."""
    import c_oracle
    N, n_obs = 8, 10
    probs = list(closed_loop_problems(N, n_obs, 6, 16, seed=77))
    B = len(probs)
    st = np.array([p[0] for p in probs]); goal = np.array([p[1] for p in probs], float)
    foot = np.array([p[2] for p in probs], np.int8); delta = np.array([p[4] for p in probs], float)
    xy, nv = lipmpc.pack_rings([p[3] for p in probs], n_obs, 5)
    args = (_dev(st, torch.float64), _dev(goal, torch.float64), _dev(foot, torch.int8), _dev(xy, torch.float64), _dev(nv, torch.int32),
            _dev(delta, torch.float64))
    res = {}
    for name, fl in (("default", 0), ("warm", lipmpc.FLAG_WARM_START), ("nopre", lipmpc.FLAG_NO_PRESOLVE)):
        P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=fl)
        o = lipmpc.BatchedLipMpc(P).plan_step_batch(*args, with_diag=True)
        torch.cuda.synchronize()
        res[name] = ({k: v.cpu().numpy() for k, v in o.items()}, c_oracle.plan_step_batch(P, st, goal, foot, xy, nv, delta, n_threads=4))
    for k in ("U", "X", "obj", "status", "iters", "active", "diag"):
        assert np.array_equal(res["warm"][0][k], res["nopre"][0][k], equal_nan=k in ("U", "X", "obj", "diag")), k
    for name in ("default", "warm"):
        g, ref = res[name]
        assert np.array_equal(g["status"], ref["status"]), name
        ok = g["status"] == 0
        assert ok.sum() > 0.8 * B and np.max(np.abs(g["U"][ok] - ref["U"][ok])) < 1e-7
        assert np.mean(g["iters"] == ref["iters"]) > 0.97, (name, np.mean(g["iters"] == ref["iters"]))
    # the two paths really differ in their interior iterates (otherwise this test proves nothing)
    assert np.mean(res["default"][0]["iters"] != res["warm"][0]["iters"]) > 0.05
    assert np.array_equal(res["warm"][1]["iters"], res["nopre"][1]["iters"])


@pytest.mark.parametrize("N,n_obs", [(8, 10), (8, 14), (6, 22), (12, 9), (16, 14), (16, 30)])
def test_crowded_robots_reach_every_solver_body(N, n_obs):
    """Which solver body a wave runs depends on how many obstacles keep a row after the presolve (1, 2, 7 or the handle's row
    slots per lane).  Robots in the middle of a ring of small obstacles -- 0 to n_obs of them within reach, a different number
    per robot -- send waves to every body of the dispatching kernel; statuses, footsteps and decisive active sets against the
    C oracle, and bit-identical answers from the kernel that keeps every row in the handle's own body."""
    import c_oracle
    from helpers import crowded_batch
    B = 256
    st, goal, foot, xy, nv = crowded_batch(N, n_obs, B, seed=7 * N + n_obs)
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
    args = (dev(st, torch.float64), dev(goal, torch.float64), dev(foot, torch.int8), dev(xy, torch.float64), dev(nv, torch.int32), None)
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5)
    out = lipmpc.BatchedLipMpc(P).plan_step_batch(*args, with_diag=True, with_working=True)
    full = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=lipmpc.FLAG_NO_PRESOLVE)).plan_step_batch(*args, with_diag=True)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in out.items()}
    f = {k: v.cpu().numpy() for k, v in full.items()}
    ref = c_oracle.plan_step_batch(P, st, goal, foot, xy, nv, None, n_threads=8)
    assert np.array_equal(g["status"], ref["status"]), (np.bincount(g["status"], minlength=6), np.bincount(ref["status"], minlength=6))
    ok = g["status"] == 0
    assert ok.sum() > B // 3
    assert np.max(np.abs(g["U"][ok] - ref["U"][ok])) < 1e-6
    info, _ = compare_active_sets(ok, g, ref)
    print("crowded", N, n_obs, info)
    assert_active_sets(f"crowded robots N={N} n_obs={n_obs}", info, 0.95)
    # ... and with every row kept: the same tight set (the rows the presolve drops are never tight)
    info_f, _ = compare_active_sets(ok & (f["status"] == 0), g, f)
    assert info_f["active_mismatch"] == 0 and info_f["active_compared_share"] >= 0.95, info_f
    # every row kept (the handle's own body, no presolve): the same optimum
    assert np.array_equal(np.isin(f["status"], (0, 4)), np.isin(g["status"], (0, 4)))
    both = ok & (f["status"] == 0)
    assert np.max(np.abs(f["U"][both] - g["U"][both])) < 1e-6
