"""CPU tests of the oracle itself (-m "not gpu"): the restatement against the reference's own
geometry outputs (hard golden), the position form against the reference's (X,U) form, the
IPM+finish against an independent Lawson-Hanson LDP solve, and the closed loop against the
trajectories recovered from the reference's committed result PDFs (soft golden)."""
import math
import os

import numpy as np
import pytest

import lipmpc_oracle as O
from helpers import (PDF_RUNS, check_pdf_bars, closed_loop_problems, load_rings, oracle_pdf_run, pdf_compare)


def test_geometry_matches_reference_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "geometry_golden.npz"))
    n_in = 0
    for q, c, eta, ins, w in zip(d["pts"], d["c"], d["eta"], d["inside"], d["which"]):
        ring = d["rings"][w][: d["nv"][w]]
        c2, e2, in2, dg = O.closest_point_and_normal(q, ring)
        assert not dg
        # the reference walks Qhull's simplex list (arbitrary endpoint order), the ring form
        # walks CCW edges: c agrees to rounding, eta to rounding / distance
        assert np.max(np.abs(c2 - c)) <= 4e-15
        dist = math.hypot(q[0] - c[0], q[1] - c[1])
        assert np.max(np.abs(e2 - eta)) <= 2e-15 + 4e-15 / dist
        assert in2 == bool(ins)          # inside flag: bit-exact, incl. points 1e-9 off an edge
        n_in += int(in2)
    assert n_in > 500


def test_theta_omega_restatement():
    P = O.Params(N=5)
    th, om = O.precompute_theta_omega(np.array([0.0, 0, 3.0, 0]), 0.0, (6, -3), P)
    # golden: first omega of Simulation1Circles (SURVEY §4 / evolution_3.pdf) = -0.49008846
    assert abs(om[0] + 0.156 * math.pi) < 1e-15
    assert abs(om[0] + 0.49008846) < 1e-8
    assert np.allclose(np.diff(th), om * P.sampling_time)


@pytest.mark.parametrize("N,n_obs", [(3, 3), (5, 3), (8, 10)])
def test_position_form_is_reference_form(N, n_obs):
    """G_q (positions) and G_U (footsteps through A_l, B_l as the reference writes them,
    HumanoidMpc.py:221-249) describe the same rows under the bijection p = T u + t0."""
    rng = np.random.default_rng(N)
    P = O.Params(N=N)
    x0 = np.array([1.0, 0.2, 2.0, -0.1])
    th, om = O.precompute_theta_omega(x0, 0.3, (10, 10), P)
    s_v = [1 if i % 2 == 0 else -1 for i in range(N + 1)]
    cs = rng.uniform(0, 5, (n_obs, 2))
    et = rng.normal(size=(n_obs, 2)); et /= np.linalg.norm(et, axis=1)[:, None]
    Gq, hq, g = O.build_qp_position_form(x0, th, om, (10, 10), s_v, cs, et, 0.1, P)
    Gu, hu, H, f, T, t0 = O.build_qp_reference_form(x0, th, om, (10, 10), s_v, cs, et, 0.1, P)
    assert np.linalg.matrix_rank(T) == 2 * N
    # rows: G_U u <= h_U  <=>  G_q (T u + t0) <= h_q
    assert np.allclose(Gq @ T, Gu, atol=1e-9 * np.abs(Gu).max())
    assert np.allclose(hq - Gq @ t0, hu, atol=1e-9 * (1 + np.abs(hu).max()))
    # cost: |T u + t0 - g|^2 has Hessian 2 T^T T — the conditioning the position form avoids
    assert np.allclose(H, 2 * T.T @ T)
    if N == 8:
        assert np.linalg.cond(H) > 1e9


@pytest.mark.parametrize("N,n_obs,ntraj,steps", [(3, 3, 4, 20), (8, 10, 3, 12)])
def test_ipm_finish_equals_independent_ldp(N, n_obs, ntraj, steps):
    """The oracle's answer is the exact minimiser: an independent active-set method
    (Lawson-Hanson NNLS on the least-distance dual) lands on the same point."""
    worst = 0.0
    for (st, goal, s0, obs, delta) in closed_loop_problems(N, n_obs, ntraj, steps, seed=N):
        P = O.Params(N=N)
        r = O.plan_step(st, goal, s0, obs, delta, P)
        assert r["status"] == O.STATUS_SOLVED
        x0 = st[:4]
        s_v = [s0 if i % 2 == 0 else -s0 for i in range(N + 1)]
        G, h, g = O.build_qp_position_form(x0, r["theta"], r["omega"], goal, s_v, r["c"], r["eta"], delta, P)
        qt = O.solve_qp_ldp_nnls(G, h, g)
        assert qt is not None
        worst = max(worst, np.max(np.abs(qt - r["q"])))
        # KKT certificate of the returned point
        assert np.min(h - G @ r["q"]) > -1e-8
    assert worst < 1e-7


def test_closed_loop_against_reference_pdf_trajectories(golden_dir):
    """Soft golden: Assets/ReportResults/Simulation1Circles{,Delta} (scenario
    simulation_1.py:85-102 / :146-160).  IPOPT (tol 1e-5) sits 1e-7 inside vertices and
    1e-4..1e-3 off in flat directions, and the loop is closed, so: first steps tight, whole run
    loose, run length within 2 steps."""
    obs = load_rings(os.path.join(golden_dir, "scenario_circles.npz"))
    pdf = np.load(os.path.join(golden_dir, "pdf_series.npz"))
    for delta, run in [(0.0, "Simulation1Circles"), (0.3, "Simulation1CirclesDelta")]:
        X, U = O.run_closed_loop((6, -3), obs, N_horizon=3, N_mpc_timesteps=300, sampling_time=0.4,
                                 init_state=(0, 0, 3, 0, 0), delta=delta, exact=False)
        ex = pdf[run + "/ev0/s0"][:, 1] + 6.0
        ey = pdf[run + "/ev0/s1"][:, 1] - 3.0
        th = pdf[run + "/ev2/s0"][:, 1]
        om = pdf[run + "/ev3/s0"][:, 1]
        assert abs(X.shape[1] - len(ex)) <= 2
        n = min(len(ex), X.shape[1])
        assert np.max(np.abs(X[0, :3] - ex[:3])) < 5e-7 and np.max(np.abs(X[2, :3] - ey[:3])) < 5e-7
        assert np.max(np.abs(X[0, :n] - ex[:n])) < 0.08 and np.max(np.abs(X[2, :n] - ey[:n])) < 0.08
        assert np.max(np.abs(X[4, :4] - th[:4])) < 1e-8 and np.max(np.abs(X[4, :6] - th[:6])) < 1e-4
        assert np.max(np.abs(U[2, :3] - om[:3])) < 1e-7
    # first exact footsteps quoted in SURVEY §4
    X, U = O.run_closed_loop((6, -3), obs, N_horizon=3, N_mpc_timesteps=3, sampling_time=0.4,
                             init_state=(0, 0, 3, 0, 0), delta=0.0, exact=True)
    assert np.allclose(U[:2, 0], [-0.0335158, 3.0743530], atol=2e-7)
    assert np.allclose(U[:2, 1], [0.0419622, 2.7924542], atol=2e-7)


@pytest.mark.parametrize("run", PDF_RUNS)
def test_closed_loop_against_every_reproducible_reference_figure(golden_dir, run):
    """Every closed loop of the reference whose inputs can be rebuilt, against the series in its committed result
    figures (tests/golden/make_pdf_pins.py): the two circle runs, the wall run (300 steps on the spot), and the three
    RRT* runs with their sub-goal lists recovered from rrt_res.pdf (3, 7 and 11 sub-goals, hand-off of
    HumanoidMPCWithRRT.py:155-181).  These figures are the only numeric outputs of the CasADi/IPOPT path that exist;
    the interior mode at IPOPT's final barrier value (helpers.IPOPT_LIKE_TOL) reproduces run lengths and first steps."""
    X, U = oracle_pdf_run(golden_dir, run)
    cmp = pdf_compare(golden_dir, run, X, U)
    print(run, X.shape[1], cmp)
    check_pdf_bars(run, X, cmp)


def test_simulation1_figure_predates_the_committed_constraints(golden_dir):
    """Assets/ReportResults/Simulation1 (simulation_1.py:33-50, BASE seed 7) is NOT a pin: its own velocity and heading
    series violate the committed manoeuvrability row (HumanoidMpc.py:204-219, 238-243) at the very first step
    (longitudinal velocity 0.50 against 0.8 - 3.6/pi * 0.49 = 0.238), so the figure was produced by an earlier
    version of the constraints.  Recorded here so that nobody spends time matching it."""
    P = np.load(os.path.join(golden_dir, "pdf_series.npz"))
    vx, vy = P["Simulation1/ev1/s0"], P["Simulation1/ev1/s1"]
    th, om = P["Simulation1/ev2/s0"], P["Simulation1/ev3/s0"]
    lon = math.cos(th[1, 1]) * vx[1, 1] + math.sin(th[1, 1]) * vy[1, 1]
    assert lon > 0.5 and 0.8 - 3.6 / math.pi * abs(om[0, 1]) < 0.24


def test_status_codes():
    P = O.Params(N=3)
    sq = np.array([[1.0, 1.0], [2.0, 1.0], [2.0, 2.0], [1.0, 2.0]])
    # robot well inside an obstacle: eta flips, the k=0 row is violated -> infeasible
    r = O.plan_step(np.array([1.5, 0, 1.4, 0, 0.0]), (5, 5), 1, [sq], 0.0, P)
    assert r["status"] == O.STATUS_INFEASIBLE
    # robot exactly on a vertex: x == c -> degenerate (reference would emit NaN, ObstaclesUtils.py:104)
    r = O.plan_step(np.array([1.0, 0, 1.0, 0, 0.0]), (5, 5), 1, [sq], 0.0, P)
    assert r["status"] == O.STATUS_DEGENERATE
    # zero-length edge
    deg = np.array([[1.0, 1.0], [1.0, 1.0], [2.0, 2.0]])
    r = O.plan_step(np.array([0.0, 0, 0.0, 0, 0.0]), (5, 5), 1, [deg], 0.0, P)
    assert r["status"] == O.STATUS_DEGENERATE
    # no obstacles at all
    r = O.plan_step(np.array([0.0, 0, 0.0, 0, 0.0]), (5, 5), 1, [], 0.0, P)
    assert r["status"] == O.STATUS_SOLVED and r["active"].shape == (27,)


def test_limit_cycle_case_converges(golden_dir):
    """A step (found among 4096 LiDAR-inferred obstacle sets, N=3, one 10-vertex hull 0.29 m away) on which plain
    Mehrotra steps fall into a 2-cycle (mu 4.26e-5 <-> 4.41e-5, steps of 0.6) and run into the iteration cap.  The
    no-progress safeguard (IPM_SLOW_*) must get it to the optimum in a normal number of iterations; without the
    safeguard the same code needs all 60."""
    d = np.load(os.path.join(golden_dir, "limit_cycle_case.npz"))
    P = O.Params(N=3)
    r = O.plan_step(d["state"], d["goal"], 1, [d["ring0"]], 0.0, P, exact=True)
    assert r["status"] == O.STATUS_SOLVED and r["iters"] <= 20, (r["status"], r["iters"])
    saved = O.IPM_SLOW_SIGMA
    try:
        O.IPM_SLOW_SIGMA = 0.0                      # safeguard off: the cycle is back
        r0 = O.plan_step(d["state"], d["goal"], 1, [d["ring0"]], 0.0, P, exact=True)
    finally:
        O.IPM_SLOW_SIGMA = saved
    assert r0["status"] == O.STATUS_MAX_ITER and r0["iters"] == P.max_iter
