"""Shared test helpers (CPU side)."""
import math

import numpy as np

import lipmpc_oracle as O


TIGHT_TOL = 1e-7       # LIPMPC_TIGHT_TOL (include/lipmpc.h): `active` bit i <=> slack_i(returned point) <= this
TIGHT_BAND = 10.0      # a problem's `active` sets are compared unless a row sits within TIGHT_BAND x (distance between the two
                       # answers) of TIGHT_TOL on either side
CERT_MARGIN = 1e-8     # working sets (multiplier-based): compared where both certificates hold with at least this margin
WORKING_DIFF_MARGIN = 1e-7   # ... and a difference ANYWHERE must sit below this certificate margin on one side
WORKING_DIFF_DU = 1e-6       # ... with the two answers this close in footstep space


def _per_problem_gap(a, b):
    d = np.abs(np.asarray(a, float) - np.asarray(b, float)).reshape(len(a), -1)
    return np.nan_to_num(np.max(d, axis=1), nan=np.inf)


def compare_active_sets(ok, g, ref):
    """"Active-constraint indices bit-exact" (BASELINE north_star), as the tests and bench.py check it.  g / ref: dicts of
    numpy arrays (GPU, oracle) with U, X, active [B,words], diag [B,8] and optionally working [B,words]; ok: the problems
    certified SOLVED on both sides.

    `active` is the PRIMAL TIGHT SET of the returned point (slack <= TIGHT_TOL).  The optimum of the strictly convex step QP
    is unique, hence so is this set; two solvers whose answers lie d apart can only disagree on a row whose slack lies within
    ~d of TIGHT_TOL at one of them.  A problem is compared unless such a row exists: tightness margin (diag[:, 4] =
    min_i |slack_i - TIGHT_TOL|) >= TIGHT_BAND x max(|dX|, |dU|) of that problem on both sides.  (X carries positions and
    velocities, the quantities the rows are written in.)  The caller asserts zero differences on the compared problems and a
    floor on their share.

    `working` (rows carrying a multiplier in the finish's certificate) is NOT unique at a degenerate vertex; it is compared
    where both certificates are decisive (diag[:, 3] >= CERT_MARGIN) and every difference, decisive or not, is reported with
    the certificate margin and footstep gap it occurs at (caller: below WORKING_DIFF_MARGIN / WORKING_DIFF_DU)."""
    dU, dX = _per_problem_gap(g["U"], ref["U"]), _per_problem_gap(g["X"], ref["X"])
    band = TIGHT_BAND * np.maximum(dU, dX)
    n_ok = max(int(ok.sum()), 1)
    compared = ok & (g["diag"][:, 4] >= band) & (ref["diag"][:, 4] >= band)
    diff = np.any(np.ascontiguousarray(g["active"]).view(np.uint64) != np.ascontiguousarray(ref["active"]).view(np.uint64), axis=1)
    info = dict(active_definition=f"primal tight set, slack <= {TIGHT_TOL:g}", active_compared=int(compared.sum()),
                active_compared_share=float(compared.sum() / n_ok), active_mismatch=int((diff & compared).sum()),
                active_mismatch_all_certified=int((diff & ok).sum()),
                active_excluded_rule=f"a row within {TIGHT_BAND:g} x max(|dX|,|dU|) of the problem from the tolerance, on either side")
    if "working" in g and "working" in ref:
        wdiff = ok & np.any(np.ascontiguousarray(g["working"]).view(np.uint64) != np.ascontiguousarray(ref["working"]).view(np.uint64), axis=1)
        cm = np.minimum(g["diag"][:, 3], ref["diag"][:, 3])
        dec = ok & (cm >= CERT_MARGIN)
        info.update(working_decisive_share=float(dec.sum() / n_ok), working_mismatch_decisive=int((wdiff & dec).sum()),
                    working_mismatch_all_certified=int(wdiff.sum()),
                    working_mismatch_max_cert_margin=float(cm[wdiff].max()) if wdiff.any() else 0.0,
                    working_mismatch_max_dU=float(dU[wdiff].max()) if wdiff.any() else 0.0)
    return info, compared


def assert_active_sets(tag, info, min_share):
    assert info["active_mismatch"] == 0, (tag, info)                       # active-constraint indices bit-exact
    assert info["active_compared_share"] >= min_share, (tag, info)
    if "working_mismatch_all_certified" in info:
        assert info["working_mismatch_max_cert_margin"] < WORKING_DIFF_MARGIN and info["working_mismatch_max_dU"] < WORKING_DIFF_DU, (tag, info)


def decisive_mask(ok, diag_a, diag_b, margin=CERT_MARGIN):
    """Problems on which the WORKING sets (rows with a positive multiplier in the finish's certificate) of two solvers must
    agree: certified on both sides (``ok``) with a decisive certificate on both sides -- diag[:, 3] = min(smallest multiplier
    on the working set, smallest slack outside it) >= margin.  Below the margin a weakly active row (multiplier ~ 0) or a
    redundant one (slack ~ 0 outside the set: dependent rows of a degenerate vertex) may legitimately sit on either side,
    which is why the canonical `active` output is the tight set instead (compare_active_sets)."""
    return ok & (diag_a[:, 3] >= margin) & (diag_b[:, 3] >= margin)


def record_parity(tag, info):
    """Observed agreement figures of a GPU-vs-oracle comparison, merged into the JSON file LIPMPC_PARITY_RECORD names
    (tools/parity_record.sh writes profiles/rNN_parity.json this way); without the variable: gpurun_out/parity_record.json."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.environ.get("LIPMPC_PARITY_RECORD") or os.path.join(root, "gpurun_out", "parity_record.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        rec = json.load(open(path)) if os.path.exists(path) else {}
        rec[tag] = info
        json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def load_rings(path, idx=None):
    d = np.load(path)
    rings, nv = d["rings"], d["nv"]
    if idx is not None:
        rings, nv = rings[idx], nv[idx]
    return [rings[i][: nv[i]] for i in range(len(nv)) if nv[i] > 0]


def synthetic_field(rng, n_obs, lo, hi):
    """Convex 3-5-gons in 1x1 boxes with centres uniform in [lo,hi]^2 (the shape of
    obstacles.py:167-194 without its rejection rules)."""
    out = []
    while len(out) < n_obs:
        c = rng.uniform(lo, hi, 2)
        pts = c + rng.uniform(-0.5, 0.5, (5, 2))
        ring = convex_ring(pts)
        if len(ring) >= 3:
            out.append(ring)
    return out


def convex_ring(pts):
    """CCW hull by Andrew's monotone chain."""
    p = sorted(map(tuple, pts))

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lo = []
    for q in p:
        while len(lo) >= 2 and cross(lo[-2], lo[-1], q) <= 0:
            lo.pop()
        lo.append(q)
    up = []
    for q in reversed(p):
        while len(up) >= 2 and cross(up[-2], up[-1], q) <= 0:
            up.pop()
        up.append(q)
    return np.array(lo[:-1] + up[:-1])


def field_features(xy, nv):
    """Shape statistics of obstacle fields [B,n_obs,v_max,2] / [B,n_obs]: vertex-count shares (3, 4, 5), polygon areas,
    nearest-neighbour distances of the polygon centres, centre coordinates, polygons per field."""
    B, n = nv.shape
    share = np.bincount(nv.ravel(), minlength=6)[3:6] / max(int((nv > 0).sum()), 1)
    area, near, cen = [], [], []
    for b in range(B):
        cs = []
        for j in range(n):
            p = xy[b, j, : nv[b, j]]
            if len(p) < 3:
                continue
            x, y = p[:, 0], p[:, 1]
            area.append(0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))
            cs.append(p.mean(0))
        cs = np.array(cs)
        cen.append(cs)
        D = np.sqrt(((cs[:, None] - cs[None]) ** 2).sum(2)) + np.eye(len(cs)) * 1e9
        near += list(D.min(1))
    return dict(vertex_share=share, area=np.array(area), nearest=np.array(near), centres=np.concatenate(cen), per_field=(nv > 0).sum(1))


def kept_rows_after_presolve(c_eta, state, delta, N):
    """Rows of the LDCBF block that stay in the problem after the presolve (oracle: presolve_ldcbf), per problem, from the
    (c, eta) rows a launch reports: obstacle j keeps its rows of the stages k with h0_j <= |eta_j| k reach_step + margin."""
    step = O.reach_step(O.Params(N=N))
    eta, c = c_eta[:, :, 2:], c_eta[:, :, :2]
    p0 = state[:, [0, 2]]
    h0 = np.einsum("bjc,bjc->bj", eta, p0[:, None, :] - c) - np.asarray(delta)[:, None]
    pres = np.any(eta != 0.0, axis=2)
    es = np.sqrt((eta ** 2).sum(2)) * step
    kept = np.zeros(h0.shape, int)
    for k in range(1, N + 1):
        kept += pres & ~(h0 > es * k + O.SCREEN_MARGIN)
    return kept.sum(1), (kept > 0).sum(1)


def crowded_batch(N, n_obs, B, seed):
    """Robots in the middle of a ring of small obstacles -- 0 to n_obs of them within reach of the horizon, a different number
    per robot, anywhere in the obstacle list -- so that after the presolve the problems of one batch need every solver body
    (1, 2, 4 / 7, 13, 25 row slots per lane).  Returns state [B,5], goal [B,2], foot [B], obs_xy [B,n_obs,5,2], obs_nv."""
    rng = np.random.default_rng(seed)
    xy = np.zeros((B, n_obs, 5, 2)); nv = np.zeros((B, n_obs), np.int32)
    st = np.zeros((B, 5)); st[:, 0] = rng.uniform(2, 8, B); st[:, 2] = rng.uniform(2, 8, B); st[:, 4] = rng.uniform(-3, 3, B)
    st[:, 3] = np.where(rng.random(B) < 0.5, 0.2, -0.2)
    foot = np.where(st[:, 3] > 0, 1, -1).astype(np.int8)
    for b in range(B):
        near = rng.integers(0, n_obs + 1)                       # obstacles within reach of the horizon
        for j in range(n_obs):
            rad = rng.uniform(0.35, 0.18 * N + 0.2) if j < near else rng.uniform(0.18 * N + 1.0, 0.18 * N + 6.0)
            ang = rng.uniform(0, 2 * np.pi)
            c = np.array([st[b, 0] + rad * np.cos(ang), st[b, 2] + rad * np.sin(ang)])
            a0 = rng.uniform(0, 2 * np.pi)
            xy[b, j, :3] = c + 0.08 * np.array([[np.cos(a0 + t), np.sin(a0 + t)] for t in (0.0, 2.1, 4.2)])     # CCW triangle
            nv[b, j] = 3
        perm = rng.permutation(n_obs)                           # the near ones anywhere in the list
        xy[b], nv[b] = xy[b, perm], nv[b, perm]
    goal = st[:, [0, 2]] + rng.uniform(-6, 6, (B, 2))
    return st, goal, foot, xy, nv


def closed_loop_problems(N, n_obs, ntraj, steps, seed=0, delta=0.0, fields=None, goal=(10.0, 10.0)):
    """Yield (state, goal, s0, obstacles, delta) along oracle closed-loop walks from the origin
    (interior iterates advance the loop), i.e. reachable walking states."""
    rng = np.random.default_rng(seed)
    P = O.Params(N=N)
    A, B = O.lip_matrices(P)
    for t in range(ntraj):
        obs = fields[t] if fields is not None else synthetic_field(rng, n_obs, 0.5, 9.5)
        st = np.zeros(5)
        for k in range(steps):
            s0 = 1 if k % 2 == 0 else -1
            yield st.copy(), goal, s0, obs, delta
            r = O.plan_step(st, goal, s0, obs, delta, P, exact=False)
            if r["status"] != O.STATUS_SOLVED:
                break
            st = np.concatenate([A @ st[:4] + B @ r["U"][0], [r["theta"][1]]])


# --------------------------------------------------------------------------------------------------------------
# the reference's committed result figures as soft pins (tests/golden/make_pdf_pins.py)
# --------------------------------------------------------------------------------------------------------------
# IPOPT (HumanoidMpc.py:99: tol 1e-5) stops on its monotone barrier schedule 0.1, 0.02, 2.8e-3, 1.5e-4, 1.8e-6 at
# mu ~ 1e-6: an interior iterate with complementarity s z ~ 1e-6.  The interior mode of the oracle / kernel stops at
# mu <= tol_interior, so 1e-6 is the setting that reproduces the reference's closed loops (run lengths of 6 of 6
# reproducible runs within 0..15 steps, whole-run position gap of the long maze run 0.024 m); the library default
# 1e-9 hugs LDCBF boundaries so closely that the loop in front of a wall (SimulationRRT-NoRRT) ends after 50 steps
# with the CoM 5e-6 inside the obstacle, where the reference walks on the spot for all 300.
IPOPT_LIKE_TOL = 1e-6

PDF_RUNS = ("Simulation1Circles", "Simulation1CirclesDelta", "SimulationRRT-NoRRT", "SimulationRRT",
            "SimulationMaze1", "SimulationMaze2")
# run -> (first steps k <= 2: position, heading; k <= 10 position; run length slack; whole-run position bound or None)
PDF_BARS = {
    "Simulation1Circles": (5e-7, 1e-8, 1e-3, 1, 0.05),
    "Simulation1CirclesDelta": (5e-7, 1e-8, 1e-3, 1, 0.05),
    "SimulationRRT-NoRRT": (5e-4, 1e-5, 5e-3, 0, 0.01),       # y is a flat direction of the cost in front of the wall
    "SimulationRRT": (2e-6, 1e-7, 1e-5, 20, None),            # 11 sub-goals: the chained loop is chaotic after ~30 steps
    "SimulationMaze1": (5e-6, 5e-7, 1e-3, 3, None),
    "SimulationMaze2": (5e-6, 5e-7, 1e-3, 3, 0.05),
    # unknown environment: the reference's sensor noise (sigma = 0.01 m per reading, unseeded) is part of its figure
    "Simulation4UnkEnv": (5e-6, 1e-7, 2e-3, 2, 0.12),
}


# run -> bars on the other panels of the figures: velocities in the robot frame (first steps k <= 2, k <= 10) and the FOOTSTEPS
# (ZMP x = U_pred[0, k]) where the figure shows them: (k <= 10, whole window); None = panel absent / not comparable
PDF_BARS_EXTRA = {
    "Simulation1Circles": (2e-6, 3e-3, 2e-3, 0.05),
    "Simulation1CirclesDelta": (2e-6, 5e-3, 3e-3, 0.05),
    "SimulationRRT-NoRRT": (2e-3, 0.06, None, None),
    "SimulationRRT": (5e-6, 2e-5, None, 0.02),           # its footstep window starts at step 23
    "SimulationMaze1": (5e-5, 5e-3, None, None),
    "SimulationMaze2": (5e-6, 1e-3, None, None),          # its CoM / ZMP panel holds four curves
    "Simulation4UnkEnv": (2e-6, 5e-3, 2e-3, 0.05),
}


def pdf_scenario(golden_dir, run):
    import os
    S = np.load(os.path.join(golden_dir, "pdf_scenarios.npz"))
    rings = [S[run + "/rings"][i][: S[run + "/nv"][i]] for i in range(len(S[run + "/nv"]))]
    sub = S[run + "/subgoals"] if run + "/subgoals" in S.files else None
    return dict(rings=rings, init=tuple(S[run + "/init"]), goal=tuple(S[run + "/goal"]), N=int(S[run + "/N"]),
                delta=float(S[run + "/delta"]), subgoals=sub)


def pdf_compare(golden_dir, run, X, U):
    """Compare a closed loop X (5,K+1), U (3,K) with the figure series of `run` at the figures' own time stamps
    (k = t / 0.4 s); returns dict of worst gaps per window and the reference's state count."""
    import os
    P = np.load(os.path.join(golden_dir, "pdf_series.npz"))
    sc = pdf_scenario(golden_dir, run)
    goal = sc["goal"]
    sigs = [("pos", 0, goal[0], "/ev0/s0"), ("pos", 2, goal[1], "/ev0/s1"), ("th", 4, 0.0, "/ev2/s0"), ("om", -1, 0.0, "/ev3/s0")]
    if run == "SimulationRRT-NoRRT":      # the X-error polyline (two straight pieces) did not survive path simplification
        sigs = [("pos", 2, goal[1], "/ev0/s0"), ("th", 4, 0.0, "/ev2/s0"), ("om", -1, 0.0, "/ev3/s0")]
    out = {}
    n_ref = 0
    for name, row, off, key in sigs:
        s = P[run + key]
        k = np.round(s[:, 0] / 0.4).astype(int)
        assert np.max(np.abs(s[:, 0] / 0.4 - k)) < 1e-5
        n_ref = max(n_ref, int(k.max()) + (2 if row == -1 else 1))
        src = U[2] if row == -1 else X[row]
        ok = k < len(src)
        d = np.abs(src[k[ok]] - off - s[ok, 1])
        kk = k[ok]
        for w, lim in (("k2", 2), ("k10", 10), ("all", 10 ** 9)):
            out[name + "_" + w] = max(out.get(name + "_" + w, 0.0), float(d[kk <= lim].max()))
    # the other panels of the figures: translational velocities in the robot frame (PlotsUtils.compute_local_velocities:
    # [[cos, sin], [-sin, cos]] (v_x, v_y) at the state's heading) and, where the panel holds exactly (CoM x, ZMP x), the
    # FOOTSTEPS themselves: ZMP x = U_pred[0, k] over the window the scripts zoom into (clipped end points are dropped)
    th = X[4]
    extra = [("vel", np.cos(th) * X[1] + np.sin(th) * X[3], "/ev1/s0"), ("vel", -np.sin(th) * X[1] + np.cos(th) * X[3], "/ev1/s1")]
    if run == "SimulationMaze1":          # this figure shows the GLOBAL velocities, as simulation_maze.py:62 plots them
        extra = [("vel", X[1], "/ev1/s0"), ("vel", X[3], "/ev1/s1")]
    if run + "/ev4/s1" in P.files and run + "/ev4/s2" not in P.files:
        extra += [("zmp", U[0], "/ev4/s1")]
    for name, src, key in extra:
        if run + key not in P.files:
            continue
        s = P[run + key]
        on_grid = np.abs(s[:, 0] / 0.4 - np.round(s[:, 0] / 0.4)) < 1e-4
        k = np.round(s[on_grid, 0] / 0.4).astype(int)
        v = s[on_grid, 1]
        ok = k < len(src)
        d, kk = np.abs(src[k[ok]] - v[ok]), k[ok]
        for w, lim in (("k2", 2), ("k10", 10), ("all", 10 ** 9)):
            if np.any(kk <= lim):
                out[name + "_" + w] = max(out.get(name + "_" + w, 0.0), float(d[kk <= lim].max()))
        if name == "zmp":
            out["zmp_first_k"], out["zmp_n"] = int(kk.min()), int(len(kk))
    out["n_ref"] = n_ref
    return out


def oracle_pdf_run(golden_dir, run, tol=IPOPT_LIKE_TOL, warm_start=False):
    sc = pdf_scenario(golden_dir, run)
    goals = sc["subgoals"] if sc["subgoals"] is not None else [sc["goal"]]
    X = U = None
    st = sc["init"]
    for g in goals:                      # HumanoidMPCWithRRT.py:155-181 hand-off; a single goal is the plain class
        xs, us = O.run_closed_loop(tuple(g), sc["rings"], N_horizon=sc["N"], N_mpc_timesteps=300, sampling_time=0.4,
                                   init_state=st, delta=sc["delta"], exact=False, params=O.Params(tol_interior=tol),
                                   warm_start=warm_start)
        st = tuple(xs[:, -1])
        X = xs if X is None else np.concatenate((X, xs), axis=1)
        U = us if U is None else np.concatenate((U, us), axis=1)
    return X, U


def check_pdf_bars(run, X, cmp):
    p2, t2, p10, nslack, pall = PDF_BARS[run]
    assert cmp["pos_k2"] <= p2 and cmp["th_k2"] <= t2, (run, cmp)
    assert cmp["pos_k10"] <= p10, (run, cmp)
    assert abs(X.shape[1] - cmp["n_ref"]) <= nslack, (run, X.shape[1], cmp["n_ref"])
    if pall is not None:
        assert cmp["pos_all"] <= pall, (run, cmp)
    v2, v10, z10, zall = PDF_BARS_EXTRA[run]
    for key, bar in (("vel_k2", v2), ("vel_k10", v10), ("zmp_k10", z10), ("zmp_all", zall)):
        if bar is not None:
            assert key in cmp and cmp[key] <= bar, (run, key, cmp.get(key), bar)


def unknown_env_scenario(golden_dir, run="Simulation4UnkEnv"):
    """The committed unknown-environment run (simulation_1.py:195-232): the true map as the raw point arrays the reference
    scans, init state, goal, horizon, LiDAR range."""
    import os
    S = np.load(os.path.join(golden_dir, "pdf_scenarios.npz"))
    env = [S[run + "/env_pts"][i][: S[run + "/env_n"][i]] for i in range(len(S[run + "/env_n"]))]
    return dict(env=env, init=tuple(S[run + "/init"]), goal=tuple(S[run + "/goal"]), N=int(S[run + "/N"]),
                lidar_range=float(S[run + "/lidar_range"]))


def oracle_unknown_env_run(golden_dir, noise_seed, run="Simulation4UnkEnv", k_max=300, tol=IPOPT_LIKE_TOL):
    """Closed loop of HumanoidMPCUnknownEnvironment on the oracle chain: per step lidar oracle (scan, noise, DBSCAN, hulls)
    -> step oracle (interior mode) -> advance; HumanoidMpc.py:380-459 with _get_list_c_and_eta from
    HumanoidMPCUnknownEnvironment.py:30-68.  noise_seed None = noiseless readings."""
    import lidar_oracle as L
    sc = unknown_env_scenario(golden_dir, run)
    P = O.Params(N=sc["N"], tol_interior=tol, sampling_time=0.4)
    A, B = O.lip_matrices(P)
    rng = np.random.default_rng(noise_seed) if noise_seed is not None else None
    tab = L.ray_table()
    st = np.array(sc["init"], float)
    X, U, last = [st.copy()], [], math.inf
    for k in range(k_max):
        if last < 0.05:                                                        # HumanoidMpc.py:392
            break
        noise = None if rng is None else L.NOISE_STD * rng.standard_normal((len(tab), 2))
        _, _, _, inferred = L.range_finder((st[0], st[2]), sc["env"], sc["lidar_range"], noise=noise, table=tab)
        r = O.plan_step(st, sc["goal"], 1 if k % 2 == 0 else -1, inferred, 0.0, P, exact=False)
        if r["status"] not in (O.STATUS_SOLVED, O.STATUS_UNCERTIFIED):
            break
        last = r["obj"]
        U.append([r["U"][0][0], r["U"][0][1], r["omega"][0]])
        st = np.concatenate([A @ st[:4] + B @ r["U"][0], [r["theta"][1]]])
        X.append(st.copy())
    return np.array(X).T, np.array(U).T
