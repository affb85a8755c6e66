"""Batched step solver: the Python face of the C ABI.  Tensors live on the GPU (torch is used
only to own device memory and streams); every call is asynchronous on torch's current stream."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib

STATUS_SOLVED, STATUS_MAX_ITER, STATUS_INFEASIBLE, STATUS_DEGENERATE, STATUS_UNCERTIFIED = 0, 1, 2, 3, 4
STATUS_SENSOR_OVERFLOW = 5     # a scan's clusters did not fit the obstacle slots: not solved (sense_plan_step, plan_step_batch_c_eta(overflow=)), robot stopped (fleet loop)
FLAG_INTERIOR = 1
FLAG_WARM_START = 2      # rollout: start every step from the previous step's shifted interior-point result
FLAG_NO_PRESOLVE = 4     # keep the LDCBF rows the leg-reach rows make redundant in the solve (include/lipmpc.h)


@dataclass
class LipMpcParams:
    """Constants of the step problem; defaults are the reference's config.yml:2-17 and
    HumanoidMpc.py:20-22,:200."""
    N: int = 3
    n_obs_max: int = 0
    v_max: int = 5
    max_iter: int = 60
    finish_rounds: int = 0      # 0 = library default (8 active-set rounds for N <= 8, else 16)
    flags: int = 0
    dt: float = 0.4
    g: float = 9.81
    h_com: float = 1.0
    alpha: float = 3.6
    l_max: tuple = (0.10, 0.10)
    l_min: tuple = (-0.1, -0.1)
    v_min: tuple = (-0.1, 0.1)
    v_max_xy: tuple = (0.8, 0.4)
    omega_max: float = 0.156 * math.pi
    ell: float = 0.05
    sampling_time: float = 0.4
    tol: float = 1e-11
    tol_interior: float = 1e-9
    k0_tol: float = 1e-5

    def to_c(self):
        p = _lib.LipmpcParamsC()
        for f in ("N", "n_obs_max", "v_max", "max_iter", "flags", "finish_rounds"):
            setattr(p, f, int(getattr(self, f)))
        for f in ("dt", "g", "h_com", "alpha", "omega_max", "ell", "sampling_time", "tol", "tol_interior", "k0_tol"):
            setattr(p, f, float(getattr(self, f)))
        for f in ("l_max", "l_min", "v_min", "v_max_xy"):
            v = getattr(self, f)
            setattr(p, f, (C.c_double * 2)(float(v[0]), float(v[1])))
        return p

    @property
    def num_rows(self):
        return 9 * self.N + (self.N + 1) * self.n_obs_max

    @property
    def active_words(self):
        return (self.num_rows + 63) // 64


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class BatchedLipMpc:
    """One handle = one (device, parameter set).  ``plan_step_batch`` solves B independent MPC
    steps; ``advance`` applies the reference's state update to the states in place."""

    def __init__(self, params: LipMpcParams, device: int | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("lipmpc needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.lib = _lib.load()
        self.params = params
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self._h = C.c_void_p()
        cp = params.to_c()
        _lib.check(self.lib.lipmpc_create(C.byref(cp), self.device_index, C.byref(self._h)), "lipmpc_create")
        # split launch (one kernel per solver body): the library says whether this handle's steps can use it
        self.auto_workspace = True
        self._ws, self._ws_cap = None, 0
        self._split_capable = int(self.lib.lipmpc_workspace_bytes(self._h, 1)) > 0

    def set_workspace(self, capacity):
        """Split launch of this handle's step solves (lipmpc_set_workspace): for 32-lane problems (N > 8) in the exact mode
        the step runs as classification -> index lists -> one kernel per solver body (each with its own register
        allocation); same optimum, statuses and active sets as the single kernel (a problem may run in another body there:
        last-bit differences).  The handle sets one up by itself for the batch sizes it sees
        (``auto_workspace``); 0 = back to the single dispatching kernel."""
        capacity = int(capacity)
        nbytes = int(self.lib.lipmpc_workspace_bytes(self._h, capacity)) if capacity > 0 else 0
        self._ws = torch.empty((nbytes // 4,), dtype=torch.int32, device=self.device) if nbytes > 0 else None
        self._ws_cap = capacity if self._ws is not None else 0
        _lib.check(self.lib.lipmpc_set_workspace(self._h, _ptr(self._ws), self._ws_cap), "lipmpc_set_workspace")

    def _ensure_workspace(self, B):
        if self.auto_workspace and B > getattr(self, "_ws_cap", 0) and self._split_capable:
            self.set_workspace(B)

    def set_schedule(self, capacity):
        """Launch order for this handle's step solves (lipmpc_set_schedule): every plan_step_batch / plan_step_batch_c_eta of
        at most ``capacity`` problems leaves each problem's cost and the order -- costliest first, like with like -- the
        next launch of the same batch size places them in.  Pays off beyond the 4096 problems the GPU holds at once,
        when consecutive launches see the same or slowly moving problems; results never depend on it.  0 = off."""
        capacity = int(capacity)
        self._sched = (torch.zeros((int(self.lib.lipmpc_schedule_words(capacity)),), dtype=torch.int32, device=self.device)
                       if capacity > 0 else None)
        _lib.check(self.lib.lipmpc_set_schedule(self._h, _ptr(self._sched), capacity), "lipmpc_set_schedule")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and C is not None:      # (C is None: interpreter shutdown)
            self.lib.lipmpc_destroy(h)
            self._h = None

    # ---- buffers --------------------------------------------------------------------------------
    def alloc_outputs(self, B, with_c_eta=False, with_diag=False, with_working=False):
        P, dev = self.params, self.device
        f64 = dict(dtype=torch.float64, device=dev)
        out = dict(
            U=torch.empty((B, P.N, 2), **f64), X=torch.empty((B, P.N + 1, 4), **f64),
            theta=torch.empty((B, P.N + 1), **f64), omega=torch.empty((B, P.N), **f64),
            obj=torch.empty((B,), **f64),
            status=torch.empty((B,), dtype=torch.int32, device=dev),
            iters=torch.empty((B,), dtype=torch.int32, device=dev),
            active=torch.empty((B, P.active_words), dtype=torch.int64, device=dev),
        )
        if with_c_eta:
            out["c_eta"] = torch.empty((B, P.n_obs_max, 4), **f64)
        if with_diag:
            out["diag"] = torch.empty((B, _lib.DIAG_WORDS), **f64)
        if with_working:
            out["working"] = torch.empty((B, P.active_words), dtype=torch.int64, device=dev)
        return out

    def _check_inputs(self, state, goal, first_foot, obs_xy, obs_nv, delta, need_obstacles=True):
        P = self.params
        B = state.shape[0]

        def need(t, shape, dtype, name):
            if t is None:
                raise ValueError(f"{name} is required")
            if tuple(t.shape) != shape or t.dtype != dtype or t.device != self.device or not t.is_contiguous():
                raise ValueError(f"{name}: expected contiguous {dtype} {shape} on {self.device}, got "
                                 f"{t.dtype} {tuple(t.shape)} on {t.device}")
        need(state, (B, 5), torch.float64, "state")
        need(goal, (B, 2), torch.float64, "goal")
        need(first_foot, (B,), torch.int8, "first_foot")
        if P.n_obs_max > 0 and need_obstacles:
            need(obs_xy, (B, P.n_obs_max, P.v_max, 2), torch.float64, "obs_xy")
            need(obs_nv, (B, P.n_obs_max), torch.int32, "obs_nv")
        if delta is not None:
            need(delta, (B,), torch.float64, "delta")
        return B

    # ---- the hot path -----------------------------------------------------------------------------
    def plan_step_batch(self, state, goal, first_foot, obs_xy=None, obs_nv=None, delta=None, out=None,
                        with_c_eta=False, with_diag=False, bounds=None, with_working=False):
        """state [B,5] (px,vx,py,vy,theta), goal [B,2], first_foot [B] int8 (+1 right / -1 left),
        obs_xy [B,n_obs_max,v_max,2] CCW rings, obs_nv [B,n_obs_max] int32, delta [B] or None,
        bounds [B,4] (V_MAX_x, V_MAX_y, ALPHA, OMEGA_MAX) per problem or None.
        Returns dict(U,X,theta,omega,obj,status,iters,active[,c_eta][,diag][,working]) of device tensors; results are
        valid once the current stream is synchronised.  ``active`` = the rows tight at the optimum (slack <= 1e-7: unique),
        ``working`` = the rows carrying a multiplier in the finish's certificate (include/lipmpc.h)."""
        B = self._check_inputs(state, goal, first_foot, obs_xy, obs_nv, delta)
        self._check_optional(bounds, (B, 4), torch.float64, "bounds")
        if out is None:
            out = self.alloc_outputs(B, with_c_eta, with_diag, with_working)
        else:
            self._check_outputs(out, B)
        self._ensure_workspace(B)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.lipmpc_plan_step_batch(
            self._h, B, _ptr(state), _ptr(goal), _ptr(first_foot), _ptr(delta), _ptr(obs_xy), _ptr(obs_nv),
            _ptr(out["U"]), _ptr(out["X"]), _ptr(out["theta"]), _ptr(out["omega"]), _ptr(out["obj"]),
            _ptr(out["status"]), _ptr(out["iters"]), _ptr(out["active"]), _ptr(out.get("working")), _ptr(out.get("c_eta")),
            _ptr(out.get("diag")), _ptr(bounds), C.c_void_p(stream))
        _lib.check(rc, "lipmpc_plan_step_batch")
        return out

    def plan_step_batch_c_eta(self, state, goal, first_foot, c_eta_in, delta=None, out=None, with_diag=False, bounds=None,
                              overflow=None, with_working=False):
        """The step with the LDCBF half-spaces given (lipmpc_plan_step_batch_c_eta): c_eta_in [B,n_obs_max,4] =
        (c_x, c_y, eta_x, eta_y) per slot, eta = (0,0) = empty slot; row j of stage k is eta_j.(p_k - c_j) - delta >= 0.
        This is what a subclass overriding the reference's _get_list_c_and_eta / _compute_single_lcbf hooks feeds.
        overflow [B] int32 or None: the flags of whoever produced the rows (LidarSensor.sense: the scan's clusters did not fit
        the obstacle slots); a flagged problem is not solved against its truncated list: status STATUS_SENSOR_OVERFLOW, NaN
        outputs (advance() leaves the robot where it is)."""
        P = self.params
        B = self._check_inputs(state, goal, first_foot, None, None, delta, need_obstacles=False)
        if (c_eta_in is None or tuple(c_eta_in.shape) != (B, P.n_obs_max, 4) or c_eta_in.dtype != torch.float64
                or c_eta_in.device != self.device or not c_eta_in.is_contiguous()):
            raise ValueError(f"c_eta_in: expected contiguous float64 {(B, P.n_obs_max, 4)} on {self.device}")
        self._check_optional(bounds, (B, 4), torch.float64, "bounds")
        self._check_optional(overflow, (B,), torch.int32, "overflow")
        if out is None:
            out = self.alloc_outputs(B, False, with_diag, with_working)
        else:
            self._check_outputs(out, B)
        self._ensure_workspace(B)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.lipmpc_plan_step_batch_c_eta(
            self._h, B, _ptr(state), _ptr(goal), _ptr(first_foot), _ptr(delta), _ptr(c_eta_in), _ptr(overflow),
            _ptr(out["U"]), _ptr(out["X"]), _ptr(out["theta"]), _ptr(out["omega"]), _ptr(out["obj"]),
            _ptr(out["status"]), _ptr(out["iters"]), _ptr(out["active"]), _ptr(out.get("working")), _ptr(out.get("diag")),
            _ptr(bounds), C.c_void_p(stream))
        _lib.check(rc, "lipmpc_plan_step_batch_c_eta")
        return out

    def _check_optional(self, t, shape, dtype, name):
        if t is not None and (tuple(t.shape) != shape or t.dtype != dtype or t.device != self.device or not t.is_contiguous()):
            raise ValueError(f"{name}: expected contiguous {dtype} {shape} on {self.device}, got {t.dtype} {tuple(t.shape)} on {t.device}")

    def _check_outputs(self, out, B, with_c_eta=False):
        """caller-supplied output buffers must have the shapes alloc_outputs gives (raw pointers go to the kernel)"""
        ref = {"U": ((B, self.params.N, 2), torch.float64), "X": ((B, self.params.N + 1, 4), torch.float64),
               "theta": ((B, self.params.N + 1), torch.float64), "omega": ((B, self.params.N), torch.float64),
               "obj": ((B,), torch.float64), "status": ((B,), torch.int32), "iters": ((B,), torch.int32),
               "active": ((B, self.params.active_words), torch.int64)}
        for k, (shape, dt) in ref.items():
            if k not in out:
                raise ValueError(f"out['{k}'] missing")
            self._check_optional(out[k], shape, dt, f"out['{k}']")
        self._check_optional(out.get("c_eta"), (B, self.params.n_obs_max, 4), torch.float64, "out['c_eta']")
        self._check_optional(out.get("diag"), (B, _lib.DIAG_WORDS), torch.float64, "out['diag']")
        self._check_optional(out.get("working"), (B, self.params.active_words), torch.int64, "out['working']")

    def advance(self, state, first_foot, out):
        """In place: state <- (A_l x + B_l U[:,0], theta[:,1]), first_foot <- -first_foot for the
        problems whose status is solved (HumanoidMpc.py:432-447)."""
        B = state.shape[0]
        self._check_optional(state, (B, 5), torch.float64, "state")
        self._check_optional(first_foot, (B,), torch.int8, "first_foot")
        self._check_outputs(out, B)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.lipmpc_advance_batch(self._h, B, _ptr(state), _ptr(first_foot), _ptr(out["U"]),
                                           _ptr(out["theta"]), _ptr(out["status"]), C.c_void_p(stream))
        _lib.check(rc, "lipmpc_advance_batch")


    def fleet_update(self, fleet, out, overflow=None, stop_obj=0.05):
        """One sample of a host-driven fleet loop after ``plan_step_batch(..., out=out)`` on the same stream:
        stop rule, stop on a failed solve, state advance, counters and the trajectory row, in one launch
        (lipmpc_fleet_update_batch).  ``fleet`` = dict(state, first_foot, walking int8, last_obj, n_steps, last_status,
        n_overflow, sample int32[1], X_pred [B,k_max+1,5], U_pred [B,k_max,3]); the device-side sample counter
        advances by one per call."""
        B, k_max = fleet["state"].shape[0], fleet["U_pred"].shape[1]
        for name, shape, dt in (("state", (B, 5), torch.float64), ("first_foot", (B,), torch.int8), ("walking", (B,), torch.int8),
                                ("last_obj", (B,), torch.float64), ("n_steps", (B,), torch.int32), ("last_status", (B,), torch.int32),
                                ("n_overflow", (B,), torch.int32), ("sample", (1,), torch.int32),
                                ("X_pred", (B, k_max + 1, 5), torch.float64), ("U_pred", (B, k_max, 3), torch.float64)):
            if name not in fleet:
                raise ValueError(f"fleet['{name}'] missing")
            self._check_optional(fleet[name], shape, dt, f"fleet['{name}']")
        self._check_outputs(out, B)
        self._check_optional(overflow, (B,), torch.int32, "overflow")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.lipmpc_fleet_update_batch(
            self._h, B, int(k_max), float(stop_obj), _ptr(fleet["state"]), _ptr(fleet["first_foot"]), _ptr(fleet["walking"]),
            _ptr(fleet["last_obj"]), _ptr(fleet["n_steps"]), _ptr(fleet["last_status"]), _ptr(fleet["n_overflow"]),
            _ptr(fleet["sample"]), _ptr(fleet["X_pred"]), _ptr(fleet["U_pred"]), _ptr(out["U"]), _ptr(out["theta"]),
            _ptr(out["omega"]), _ptr(out["obj"]), _ptr(out["status"]), _ptr(overflow), C.c_void_p(stream))
        _lib.check(rc, "lipmpc_fleet_update_batch")

    def rollout(self, state0, goal, first_foot, obs_xy=None, obs_nv=None, delta=None, k_max=100, mpc_step=1,
                stop_obj=0.05, bounds=None):
        """Closed loop on the device (HumanoidMpc.py:345-459) for B robots: returns dict(X_pred [B,k_max+1,5],
        U_pred [B,k_max,3], n_steps [B], last_status [B], total_iters [B]); rows beyond n_steps are undefined."""
        B = self._check_inputs(state0, goal, first_foot, obs_xy, obs_nv, delta)
        self._check_optional(bounds, (B, 4), torch.float64, "bounds")
        if int(k_max) < 1 or int(mpc_step) < 1:
            raise ValueError("k_max and mpc_step must be positive")
        dev = self.device
        out = dict(X_pred=torch.empty((B, k_max + 1, 5), dtype=torch.float64, device=dev),
                   U_pred=torch.empty((B, k_max, 3), dtype=torch.float64, device=dev),
                   n_steps=torch.empty((B,), dtype=torch.int32, device=dev),
                   last_status=torch.empty((B,), dtype=torch.int32, device=dev),
                   total_iters=torch.empty((B,), dtype=torch.int32, device=dev))
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.lipmpc_rollout_batch(self._h, B, int(k_max), int(mpc_step), float(stop_obj), _ptr(state0), _ptr(goal),
                                           _ptr(first_foot), _ptr(delta), _ptr(obs_xy), _ptr(obs_nv), _ptr(out["X_pred"]),
                                           _ptr(out["U_pred"]), _ptr(out["n_steps"]), _ptr(out["last_status"]),
                                           _ptr(out["total_iters"]), _ptr(bounds), C.c_void_p(stream))
        _lib.check(rc, "lipmpc_rollout_batch")
        return out

    def rollout_subgoals(self, state0, sub_goals, n_sub, first_foot, obs_xy=None, obs_nv=None, delta=None, k_max=100,
                         mpc_step=1, stop_obj=0.05, bounds=None):
        """Sub-goal sequencing for B robots (the hand-off of HumanoidMPCWithRRT.py:155-181): robot b walks to
        sub_goals[b, 0], then from where it stopped to sub_goals[b, 1], ... for n_sub[b] segments; every segment is
        a fresh closed loop (foot schedule restarts at first_foot[b], own k_max budget) kept with the reference's
        truncation, so a segment that uses all k_max samples hands over its last-but-one state.  A robot whose
        solve fails stops there.  One rollout launch per segment over the robots still walking.
        Returns dict(X_pred [B,S,k_max+1,5], U_pred [B,S,k_max,3], n_kept [B,S] kept inputs per segment
        (kept states = n_kept+1; -1 = segment not run), last_status [B], final_state [B,5])."""
        B = self._check_inputs(state0, sub_goals[:, 0].contiguous(), first_foot, obs_xy, obs_nv, delta)
        S = sub_goals.shape[1]
        dev = self.device
        out = dict(X_pred=torch.zeros((B, S, k_max + 1, 5), dtype=torch.float64, device=dev),
                   U_pred=torch.zeros((B, S, k_max, 3), dtype=torch.float64, device=dev),
                   n_kept=torch.full((B, S), -1, dtype=torch.int32, device=dev),
                   last_status=torch.zeros((B,), dtype=torch.int32, device=dev),
                   final_state=state0.clone())
        alive = torch.ones((B,), dtype=torch.bool, device=dev)
        sel = lambda t, i: None if t is None else t.index_select(0, i).contiguous()
        for s in range(S):
            idx = torch.nonzero(alive & (n_sub.to(dev) > s)).flatten()
            if idx.numel() == 0:
                break
            ro = self.rollout(sel(out["final_state"], idx), sel(sub_goals[:, s], idx), sel(first_foot, idx),
                              sel(obs_xy, idx), sel(obs_nv, idx), sel(delta, idx), k_max=k_max, mpc_step=mpc_step,
                              stop_obj=stop_obj, bounds=sel(bounds, idx))
            n = ro["n_steps"].to(torch.int64)
            kept = torch.where(n < k_max, n, torch.full_like(n, k_max - 1))        # HumanoidMpc.py:457-459
            out["X_pred"][idx, s] = ro["X_pred"]
            out["U_pred"][idx, s] = ro["U_pred"]
            out["n_kept"][idx, s] = kept.to(torch.int32)
            out["last_status"][idx] = ro["last_status"]
            out["final_state"][idx] = ro["X_pred"][torch.arange(idx.numel(), device=dev), kept]
            ok = (ro["last_status"] == STATUS_SOLVED) | (ro["last_status"] == STATUS_UNCERTIFIED)
            alive[idx] = ok
        return out


def unpack_active(active_words: np.ndarray, num_rows: int) -> np.ndarray:
    """[B,words] int64/uint64 -> [B,num_rows] bool in canonical row order."""
    w = np.ascontiguousarray(active_words).view(np.uint64)
    bits = (w[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)
    return bits.reshape(w.shape[0], -1)[:, :num_rows].astype(bool)


def pack_rings(obstacle_sets, n_obs_max, v_max):
    """list (per problem) of lists of (V,2) CCW rings -> (obs_xy [B,n_obs_max,v_max,2], obs_nv [B,n_obs_max])."""
    B = len(obstacle_sets)
    xy = np.zeros((B, n_obs_max, v_max, 2))
    nv = np.zeros((B, n_obs_max), np.int32)
    for b, rings in enumerate(obstacle_sets):
        if len(rings) > n_obs_max:
            raise ValueError(f"problem {b}: {len(rings)} obstacles > n_obs_max={n_obs_max}")
        for j, r in enumerate(rings):
            r = np.asarray(r, float)
            if r.shape[0] > v_max:
                raise ValueError(f"problem {b} obstacle {j}: {r.shape[0]} vertices > v_max={v_max}")
            xy[b, j, : r.shape[0]] = r
            nv[b, j] = r.shape[0]
    return xy, nv
