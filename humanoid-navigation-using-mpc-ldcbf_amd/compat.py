"""Drop-in mirrors of the reference's controller classes, running every MPC step on the GPU.

``HumanoidMPC`` keeps the constructor and ``run_simulation`` signature, defaults, return shapes
and truncation behaviour of HumanoidNavigation/MPC/HumanoidMpc.py:50-52, 345-347, 457-459, 494;
``HumanoidMPCCustomLCBF`` adds ``distance_from_obstacles`` (HumanoidMPCCustomLCBF.py:15-28).
Plotting/animation (HumanoidAnimationUtils) is out of scope: the plotting arguments are accepted
and ignored, and the third return value is ``initial_animator`` (None by default).
"""
from __future__ import annotations

import math
from typing import Union

import numpy as np
import torch

from .solver import (BatchedLipMpc, LipMpcParams, FLAG_INTERIOR, FLAG_WARM_START, STATUS_SOLVED, STATUS_UNCERTIFIED,
                     pack_rings)

DELTA_T = 0.4  # config.yml:2


def _ring_of(obstacle):
    """ConvexHull -> CCW vertex ring ``points[vertices]`` (ObstaclesUtils.py:55); arrays pass through."""
    if hasattr(obstacle, "vertices") and hasattr(obstacle, "points"):
        return np.asarray(obstacle.points, float)[obstacle.vertices]
    return np.asarray(obstacle, float)


class HumanoidMPC:
    def __init__(self, goal, obstacles, N_horizon=3, N_mpc_timesteps=100, sampling_time=1e-3,
                 init_state: Union[np.ndarray, tuple] = np.array([0, 0, 0, 0, 0]),
                 start_with_right_foot: bool = True, verbosity: int = 1, *, exact: bool = False,
                 interior_tol: float = 1e-6, warm_start: bool = False, device: int | None = None):
        # HumanoidMpc.py:66
        assert DELTA_T % sampling_time <= 1e-8, \
            "The sampling time must be lower than and divisible by the duration of the step."
        self.N_horizon = N_horizon
        self.N_simul = N_mpc_timesteps
        self.sampling_time = sampling_time
        self.mpc_step = int(DELTA_T / sampling_time) or 1          # :74-75
        self.num_inputs = self.mpc_step * self.N_simul              # :78
        self.start_with_right_foot = start_with_right_foot
        self.verbosity = verbosity
        self.goal = goal
        self.obstacles = obstacles
        assert self.goal is not None and self.obstacles is not None  # :91
        if isinstance(init_state, tuple):
            init_state = np.array(init_state)
        init_state = np.asarray(init_state, float)
        assert init_state.shape[0] == 5, "The initial state must be a vector with 5 components."
        self.init_state = init_state
        # s_v, HumanoidMpc.py:104-108
        base = 0 if start_with_right_foot else 1
        self.s_v = [1 if i % 2 == base else -1 for i in range(self.num_inputs + N_horizon + 1)]
        self.precomputed_theta = None
        self.precomputed_omega = None
        self.distance_from_obstacles = getattr(self, "distance_from_obstacles", 0.0)
        self._exact = exact
        # Stop tolerance of the interior mode (exact=False).  1e-6 is where the reference's IPOPT (tol 1e-5,
        # HumanoidMpc.py:99) ends its barrier schedule; with it the closed loops reproduce the reference's committed
        # result figures (tests/golden/make_pdf_pins.py: run lengths, first steps).  Tighter values hug LDCBF
        # boundaries more closely than IPOPT does and end runs in front of walls early.
        self._interior_tol = interior_tol
        # warm_start=True: every MPC step of the on-device loop starts from the previous step's shifted result (the
        # reference seeds its next solve with the shifted prediction, HumanoidMpc.py:450-455): same optimum, 15-30 % fewer
        # iterations.  Off by default: the interior iterate a step stops at depends on where it started, and the closed
        # loops that reproduce the reference's figures (tests/golden/make_pdf_pins.py) are the cold-started ones -- the
        # 7-sub-goal maze run ends in an infeasible corner 90 steps early when warm-started.
        self._warm_start = warm_start
        self._device = device
        self._solver = None
        self._ce_solver = None
        self._rings = None
        self.last_status = None

    # -- hooks kept from the reference ------------------------------------------------------------
    def _get_obstacle_rings(self, x_k: float, y_k: float):
        """Obstacles seen at this step as CCW rings (override for sensed environments)."""
        if self._rings is None:
            self._rings = [_ring_of(o) for o in self.obstacles]
        return self._rings

    def _make_solver(self, n_obs, v_max):
        p = LipMpcParams(N=self.N_horizon, n_obs_max=n_obs, v_max=max(3, v_max),
                         sampling_time=self.sampling_time,
                         flags=(0 if self._exact else FLAG_INTERIOR) | (FLAG_WARM_START if self._warm_start else 0),
                         tol_interior=self._interior_tol)
        return BatchedLipMpc(p, self._device)

    def _plan_rings(self, state5, s0):
        rings = self._get_obstacle_rings(state5[0], state5[2])
        n_obs = len(rings)
        v_max = max([3] + [len(r) for r in rings])
        if self._solver is None or self._solver.params.n_obs_max != n_obs or self._solver.params.v_max < v_max:
            self._solver = self._make_solver(n_obs, v_max)
        sv = self._solver
        dev = sv.device
        xy, nv = pack_rings([rings], n_obs, sv.params.v_max)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        out = sv.plan_step_batch(
            t(state5[None, :], torch.float64), t(np.asarray(self.goal, float)[None, :], torch.float64),
            t(np.array([s0], np.int8), torch.int8),
            t(xy, torch.float64) if n_obs else None, t(nv, torch.int32) if n_obs else None,
            t(np.array([self.distance_from_obstacles], float), torch.float64), with_c_eta=n_obs > 0)
        torch.cuda.synchronize(dev)
        return {k: v[0].cpu().numpy() for k, v in out.items()}

    def _get_list_c_and_eta(self, x_k: float, y_k: float):
        """HumanoidMpc.py:296-319 — (list_c, list_eta) of (2,1) arrays at the given CoM position (computed by the
        kernel's geometry front end).  A subclass may override it, as the reference's unknown-environment variant does
        (HumanoidMPCUnknownEnvironment.py:30-68): the step is then solved against the half-spaces it returns."""
        st = np.array([x_k, 0.0, y_k, 0.0, 0.0])
        r = self._plan_rings(st, 1)
        if "c_eta" not in r:
            return [], []
        return ([ce[:2].reshape(2, 1) for ce in r["c_eta"]], [ce[2:].reshape(2, 1) for ce in r["c_eta"]])

    def _compute_single_lcbf(self, x, eta, c):
        """HumanoidMpc.py:252-261 — h(x) = eta^T (x - c), numeric here (x, eta, c are (2,1) arrays).  A subclass may
        override it with any function AFFINE in x (the reference's own override subtracts a margin,
        HumanoidMPCCustomLCBF.py:30-31): the solver recovers the half-space from three evaluations."""
        return float(np.asarray(eta, float).reshape(2) @ (np.asarray(x, float).reshape(2) - np.asarray(c, float).reshape(2)))

    def _hooks_overridden(self):
        t = type(self)
        return (t._get_list_c_and_eta is not HumanoidMPC._get_list_c_and_eta
                or t._compute_single_lcbf not in (HumanoidMPC._compute_single_lcbf, HumanoidMPCCustomLCBF._compute_single_lcbf))

    def _plan(self, state5, s0, lists=None):
        """One MPC step.  Stock hooks: rings -> kernel front end.  Overridden hooks: the subclass's (c, eta) lists and
        its h(x), turned into data for lipmpc_plan_step_batch_c_eta (``lists``: what the hook already returned for this
        sample -- the closed loop calls it once per sample, before its stop test, as the reference does)."""
        if not self._hooks_overridden():
            return self._plan_rings(state5, s0)
        list_c, list_eta = lists if lists is not None else self._get_list_c_and_eta(float(state5[0]), float(state5[2]))
        rows = []
        lcbf_stock = type(self)._compute_single_lcbf in (HumanoidMPC._compute_single_lcbf, HumanoidMPCCustomLCBF._compute_single_lcbf)
        delta = None
        if lcbf_stock:
            # h(x) = eta.(x - c) - distance_from_obstacles: the hook's (c, eta) go to the solver as they are
            rows = [np.concatenate([np.asarray(c, float).reshape(2), np.asarray(eta, float).reshape(2)]) for c, eta in zip(list_c, list_eta)]
            delta = float(self.distance_from_obstacles)
        else:
            e0, e1, e2 = np.zeros((2, 1)), np.array([[1.0], [0.0]]), np.array([[0.0], [1.0]])
            for c, eta in zip(list_c, list_eta):
                h0 = float(self._compute_single_lcbf(e0, eta, c))
                a = np.array([float(self._compute_single_lcbf(e1, eta, c)) - h0, float(self._compute_single_lcbf(e2, eta, c)) - h0])
                n2 = float(a @ a)
                if n2 > 0.0:                      # h(x) = a.x + h0 = a.(x - c') with c' = -h0 a / |a|^2
                    rows.append(np.concatenate([-h0 * a / n2, a]))
                elif n2 != n2:                    # NaN from the hook (degenerate geometry): let the solver report it
                    rows.append(np.array([0.0, 0.0, np.nan, np.nan]))
        n_obs = len(rows)
        if self._ce_solver is None or self._ce_solver.params.n_obs_max != n_obs:
            p = LipMpcParams(N=self.N_horizon, n_obs_max=n_obs, v_max=3, sampling_time=self.sampling_time,
                             flags=0 if self._exact else FLAG_INTERIOR, tol_interior=self._interior_tol)
            self._ce_solver = BatchedLipMpc(p, self._device)
        sv = self._ce_solver
        dev = sv.device
        t = lambda arr, dt: torch.as_tensor(np.ascontiguousarray(arr), dtype=dt, device=dev)
        ce = t(np.array(rows).reshape(1, n_obs, 4) if n_obs else np.zeros((1, 0, 4)), torch.float64)
        out = sv.plan_step_batch_c_eta(t(state5[None, :], torch.float64), t(np.asarray(self.goal, float)[None, :], torch.float64),
                                       t(np.array([s0], np.int8), torch.int8), ce,
                                       None if delta is None else t(np.array([delta]), torch.float64))
        torch.cuda.synchronize(dev)
        return {k: v[0].cpu().numpy() for k, v in out.items()}

    # -- the closed loop (HumanoidMpc.py:345-494) ------------------------------------------------
    def run_simulation(self, path_to_gif: str = None, make_fast_plot: bool = True, plot_animation: bool = False,
                       fill_animator: bool = True, initial_animator=None):
        if type(self)._get_obstacle_rings is HumanoidMPC._get_obstacle_rings and not self._hooks_overridden():
            return self._run_on_device(initial_animator)       # static obstacles, stock hooks: whole loop in one launch
        return self._run_stepwise(initial_animator)            # sensed / changing obstacles: one launch per sample

    def _run_on_device(self, initial_animator):
        rings = self._get_obstacle_rings(self.init_state[0], self.init_state[2])
        n_obs = len(rings)
        v_max = max([3] + [len(r) for r in rings])
        if self._solver is None or self._solver.params.n_obs_max != n_obs or self._solver.params.v_max < v_max:
            self._solver = self._make_solver(n_obs, v_max)
        sv = self._solver
        dev = sv.device
        xy, nv = pack_rings([rings], n_obs, sv.params.v_max)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        out = sv.rollout(t(self.init_state[None, :], torch.float64), t(np.asarray(self.goal, float)[None, :], torch.float64),
                         t(np.array([self.s_v[0]], np.int8), torch.int8),
                         t(xy, torch.float64) if n_obs else None, t(nv, torch.int32) if n_obs else None,
                         t(np.array([self.distance_from_obstacles], float), torch.float64),
                         k_max=self.num_inputs, mpc_step=self.mpc_step)
        torch.cuda.synchronize(dev)
        n = int(out["n_steps"][0])
        self.last_status = int(out["last_status"][0])
        if self.last_status not in (STATUS_SOLVED, STATUS_UNCERTIFIED) and self.verbosity > 0:
            print(f"===== ERROR ({n}) ===== solver status {self.last_status}")       # HumanoidMpc.py:419-429
        X = out["X_pred"][0].cpu().numpy().T
        U = out["U_pred"][0].cpu().numpy().T
        # the reference's truncation X[:, :k+1], U[:, :k] with k the last loop index (:457-459): a run that
        # uses every sample loses its final state and input, exactly as there
        k = n if n < self.num_inputs else self.num_inputs - 1
        return X[:, :k + 1].copy(), U[:, :k].copy(), initial_animator

    def _run_stepwise(self, initial_animator):
        X_pred = np.zeros((5, self.num_inputs + 1))
        U_pred = np.zeros((3, self.num_inputs))
        X_pred[:, 0] = self.init_state
        last_obj = float("inf")
        u_keep = np.zeros(2)
        ch, sh, beta = self._lip()
        k = 0
        hooked = self._hooks_overridden()
        for k in range(self.num_inputs):
            is_mpc = k % self.mpc_step == 0
            st = X_pred[:, k].copy()
            # The reference assembles the LDCBF constraints on EVERY sample, BEFORE its stop test (HumanoidMpc.py:387 vs :392:
            # a subclass's _get_list_c_and_eta runs, its scan lists grow by one entry also on the sample the run stops at),
            # but solves only on MPC samples (:415-417)
            lists = self._get_list_c_and_eta(float(st[0]), float(st[2])) if hooked else None
            if last_obj < 0.05:                                         # :392-393
                break
            step_number = math.floor(k / self.mpc_step)
            if is_mpc:
                r = self._plan(st, self.s_v[step_number], lists)
            else:
                # the other samples advance the heading with the theta / omega recurrence alone (:137-160, 443-447)
                th, om = self._theta_omega(st)
                r = {"theta": th, "omega": om}
            self.precomputed_theta, self.precomputed_omega = r["theta"], r["omega"]
            if is_mpc:
                self.last_status = int(r["status"])
                if self.last_status not in (STATUS_SOLVED, STATUS_UNCERTIFIED):
                    if self.verbosity > 0:
                        print(f"===== ERROR ({k}) ===== solver status {self.last_status}")  # :419-429
                    break
                last_obj = float(r["obj"])
                u_keep = r["U"][0]
            U_pred[:2, k] = u_keep                                      # :432-433
            U_pred[2, k] = r["omega"][0]
            if is_mpc:                                                  # :439-447
                x = st[:4]
                X_pred[0, k + 1] = ch * x[0] + sh / beta * x[1] + (1 - ch) * u_keep[0]
                X_pred[1, k + 1] = beta * sh * x[0] + ch * x[1] - beta * sh * u_keep[0]
                X_pred[2, k + 1] = ch * x[2] + sh / beta * x[3] + (1 - ch) * u_keep[1]
                X_pred[3, k + 1] = beta * sh * x[2] + ch * x[3] - beta * sh * u_keep[1]
            else:
                X_pred[:4, k + 1] = st[:4]
            X_pred[4, k + 1] = r["theta"][1]
        return X_pred[:, :k + 1], U_pred[:, :k], initial_animator      # :457-459, 494

    def _theta_omega(self, state5):
        """HumanoidMpc.py:137-160 on the host (samples between two MPC steps): omega_k = clip(atan2(goal - p_0) - theta_k,
        -OMEGA_MAX, OMEGA_MAX), theta_{k+1} = theta_k + omega_k * sampling_time -- the recurrence the kernel's front end runs."""
        om_max = 0.156 * math.pi                                    # HumanoidMpc.py:21-22
        psi = math.atan2(float(self.goal[1]) - float(state5[2]), float(self.goal[0]) - float(state5[0]))
        th, om = [float(state5[4])], []
        for _ in range(self.N_horizon):
            w = min(max(psi - th[-1], -om_max), om_max)
            om.append(w)
            th.append(th[-1] + w * self.sampling_time)
        return np.array(th), np.array(om)

    @staticmethod
    def _lip():
        beta = math.sqrt(9.81 / 1.0)
        return math.cosh(beta * DELTA_T), math.sinh(beta * DELTA_T), beta


class HumanoidMPCCustomLCBF(HumanoidMPC):
    """LDCBF with a safety margin: h = eta^T (x - c) - delta (HumanoidMPCCustomLCBF.py:30-31)."""

    def __init__(self, goal, obstacles, N_horizon=3, N_mpc_timesteps=100, sampling_time=1e-3,
                 init_state=None, start_with_right_foot: bool = True, verbosity: int = 1,
                 distance_from_obstacles: float = 0.0, **kw):
        assert distance_from_obstacles >= 0.0, "distance_from_obstacles must be non-negative"
        self.distance_from_obstacles = distance_from_obstacles
        super().__init__(goal, obstacles, N_horizon, N_mpc_timesteps, sampling_time,
                         np.zeros(5) if init_state is None else init_state, start_with_right_foot, verbosity, **kw)

    def _compute_single_lcbf(self, x, eta, c):
        """HumanoidMPCCustomLCBF.py:30-31 (the solver takes the margin as its `delta` input on the fast path)."""
        return HumanoidMPC._compute_single_lcbf(self, x, eta, c) - self.distance_from_obstacles


class HumanoidMPCWithRRT(HumanoidMPC):
    """Reach the goal through a sequence of sub-goals (HumanoidMPCVariants/HumanoidMPCWithRRT.py:15-183).

    The reference obtains the sub-goals from ``rrtplanner.RRTStar`` on an occupancy grid (:98-138) — sequential,
    CPU-side planning that is outside the accelerated path (and ``rrtplanner`` is not installed here).  What belongs
    to the MPC path is the hand-off (:155-181), mirrored exactly: one fresh ``HumanoidMPC`` per sub-goal with the
    parent's horizon / sample count / sampling time / first foot, started from the last state of the previous run,
    X_pred / U_pred concatenated as they come (the hand-off state appears twice, as there).  Like the reference
    (:155) the first run starts from (0, 0, 0, 0, 0) whatever ``init_state`` says unless ``honour_init_state=True``.

    Sub-goals come from ``sub_goals`` ([S,2], the last one normally the goal), or from ``planner(self) -> [S,2]``.
    With neither, ``run_simulation`` raises ImportError: there is no built-in planner.
    """

    def __init__(self, goal, obstacles, N_horizon=3, N_mpc_timesteps=100, sampling_time=1e-3,
                 init_state: Union[np.ndarray, tuple] = np.array([0, 0, 0, 0, 0]),
                 start_with_right_foot: bool = True, verbosity: int = 1, *, sub_goals=None, planner=None,
                 honour_init_state: bool = False, **kw):
        super().__init__(goal, obstacles, N_horizon, N_mpc_timesteps, sampling_time, init_state,
                         start_with_right_foot, verbosity, **kw)
        self.sub_goals = None if sub_goals is None else np.asarray(sub_goals, float).reshape(-1, 2)
        self.planner = planner
        self.honour_init_state = honour_init_state
        self._kw = kw

    def run_simulation(self, path_to_gif: str = None, make_fast_plot: bool = True, plot_animation: bool = False,
                       fill_animator: bool = True, initial_animator=None, visualize_rrt_path: bool = False,
                       path_to_rrt_pdf: str = None):
        if self.sub_goals is not None:
            sub_goals = self.sub_goals
        elif self.planner is not None:
            sub_goals = np.asarray(self.planner(self), float).reshape(-1, 2)
        else:
            raise ImportError("HumanoidMPCWithRRT: no planner available (rrtplanner is not part of this library); "
                              "pass sub_goals=[[x, y], ...] or planner=callable")
        X_glob, U_glob = None, None
        start_state = tuple(self.init_state) if self.honour_init_state else (0, 0, 0, 0, 0)
        animator = initial_animator
        for sub_goal in sub_goals:
            cur = HumanoidMPC(goal=sub_goal, init_state=start_state, obstacles=self.obstacles,
                              N_horizon=self.N_horizon, N_mpc_timesteps=self.N_simul, sampling_time=self.sampling_time,
                              start_with_right_foot=self.start_with_right_foot, verbosity=self.verbosity, **self._kw)
            cur.distance_from_obstacles = self.distance_from_obstacles
            cur._solver = self._solver                       # same obstacle set: keep the handle
            X, U, animator = cur.run_simulation(path_to_gif, False, False, fill_animator, animator)
            self._solver, self.last_status = cur._solver, cur.last_status
            start_state = tuple(X[:, -1])                   # :177
            X_glob = X if X_glob is None else np.concatenate((X_glob, X), axis=1)      # :179-180
            U_glob = U if U_glob is None else np.concatenate((U_glob, U), axis=1)
        return X_glob, U_glob, animator
