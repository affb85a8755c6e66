"""LiDAR front end of the unknown-environment variant (BASELINE config 5) — host wrapper of
lipmpc_lidar_sense_batch and the drop-in HumanoidMPCUnknownEnvironment class
(HumanoidNavigation/MPC/HumanoidMPCVariants/HumanoidMPCUnknownEnvironment.py:13-68)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .compat import HumanoidMPC, _ring_of
from .solver import _ptr

NOISE_STD = 0.01           # range_finder_wth_polygons_dbscan.py:163
DBSCAN_EPS = 0.3           # :100
DBSCAN_MIN_SAMPLES = 3     # :100


def ray_table(resolution=360):
    """(cos, sin) of i * 2 pi / resolution through math.cos / math.sin, as the reference computes its rays (:28-36)."""
    step = 2 * math.pi / resolution
    return np.array([[math.cos(i * step), math.sin(i * step)] for i in range(resolution)])


class LidarSensor:
    """Batched range_finder(): scan -> noise -> DBSCAN -> hulls, one wavefront per robot, rings in the layout
    BatchedLipMpc.plan_step_batch consumes."""

    def __init__(self, env_rings, lidar_range=3.0, resolution=360, n_obs_max=12, v_max=32, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("lipmpc needs a HIP device; there is no CPU path")
        self.lib = _lib.load()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.lidar_range, self.resolution, self.n_obs_max, self.v_max = float(lidar_range), int(resolution), n_obs_max, v_max
        rings = [np.asarray(r, float) for r in env_rings]
        self.n_env = len(rings)
        self.v_env = max([1] + [len(r) for r in rings])
        xy = np.zeros((max(self.n_env, 1), self.v_env, 2)); nv = np.zeros(max(self.n_env, 1), np.int32)
        for j, r in enumerate(rings):
            xy[j, :len(r)] = r; nv[j] = len(r)
        self.env_xy = torch.as_tensor(xy, device=self.device)
        self.env_nv = torch.as_tensor(nv, device=self.device)
        self.table = torch.as_tensor(ray_table(self.resolution), device=self.device)

    def alloc_outputs(self, B, with_debug=False, rings=True, c_eta=False):
        """Output buffers of ``sense``: rings (obs_xy, obs_nv) and / or the assembled half-spaces c_eta [B,n_obs_max,4]."""
        dev = self.device
        out = dict(n_inferred=torch.zeros((B,), dtype=torch.int32, device=dev),
                   overflow=torch.zeros((B,), dtype=torch.int32, device=dev))
        if rings:
            out["obs_xy"] = torch.zeros((B, self.n_obs_max, self.v_max, 2), dtype=torch.float64, device=dev)
            out["obs_nv"] = torch.zeros((B, self.n_obs_max), dtype=torch.int32, device=dev)
        if c_eta:
            out["c_eta"] = torch.zeros((B, self.n_obs_max, 4), dtype=torch.float64, device=dev)
        if with_debug:
            out["hits"] = torch.empty((B, self.resolution, 2), dtype=torch.float64, device=dev)
            out["labels"] = torch.empty((B, self.resolution), dtype=torch.int32, device=dev)
        return out

    def make_schedule(self, B):
        """An order buffer for ``sense(..., schedule=)`` (lipmpc_lidar_c_eta_batch): scratch in which the call ranks its B
        robots by an estimate of their reading counts and deals their scans out so that every SIMD gets a heavy one with light ones.  Nothing carries
        over between calls; results do not depend on it."""
        return torch.zeros((int(self.lib.lipmpc_lidar_schedule_words(B)),), dtype=torch.int32, device=self.device)

    def sense(self, state, noise=None, with_debug=False, out=None, env_xy=None, env_nv=None, c_eta=False, rings=True,
              schedule="auto"):
        """state [B,5] device tensor; noise [B,resolution,2] or None -> dict(n_inferred, overflow[, obs_xy, obs_nv][, c_eta]
        [, hits, labels]).  ``c_eta=True``: the constraint assembly runs in the same launch (lipmpc_lidar_c_eta_batch) and
        the dict carries c_eta [B,n_obs_max,4] = (c, eta) of every inferred hull at the robot's CoM -- what
        ``BatchedLipMpc.plan_step_batch_c_eta`` solves against; with ``rings=False`` the hulls never leave the kernel.
        ``schedule``: a buffer of ``make_schedule(B)``, None (robots scanned in index order), or "auto" (default): with
        ``c_eta=True`` and more than one round of waves the sensor keeps one order buffer per (batch size, current stream) and
        every scan first ranks its robots (estimated reading counts -> launch positions: two small kernels inside the call).  Any
        order gives the same results -- but scans that share a BUFFER must be ordered on one stream (the order kernel of one
        launch rewrites what another launch reads; a torn order would scan some robots twice and others not at all): a buffer
        of ``make_schedule`` handed to launches on two streams, or to two graphs replayed concurrently, is a caller's bug.
        Vertex slots beyond obs_nv keep whatever an earlier call left there when ``out`` is reused.
        ``env_xy`` [B,n_env,v_env,2] / ``env_nv`` [B,n_env] (device tensors): one true map PER ROBOT instead of the
        sensor's shared map (env_shared = 0 of the C ABI)."""
        B, dev = state.shape[0], self.device
        if out is None:
            out = self.alloc_outputs(B, with_debug, rings=rings or not c_eta, c_eta=c_eta)
        want_ce = "c_eta" in out
        for name, shape, dt in (("obs_xy", (B, self.n_obs_max, self.v_max, 2), torch.float64), ("obs_nv", (B, self.n_obs_max), torch.int32),
                                ("c_eta", (B, self.n_obs_max, 4), torch.float64), ("n_inferred", (B,), torch.int32),
                                ("overflow", (B,), torch.int32), ("hits", (B, self.resolution, 2), torch.float64),
                                ("labels", (B, self.resolution), torch.int32)):
            t = out.get(name)
            if t is not None and (tuple(t.shape) != shape or t.dtype != dt or t.device != dev or not t.is_contiguous()):
                raise ValueError(f"out['{name}']: expected contiguous {dt} {shape} on {dev}")
        if (state.dtype != torch.float64 or state.dim() != 2 or state.shape[1] != 5 or state.device != dev or not state.is_contiguous()):
            raise ValueError(f"state: expected contiguous float64 [B,5] on {dev}")
        if noise is not None and (tuple(noise.shape) != (B, self.resolution, 2) or noise.dtype != torch.float64
                                  or noise.device != dev or not noise.is_contiguous()):
            raise ValueError(f"noise: expected contiguous float64 {(B, self.resolution, 2)} on {dev}")
        stream = torch.cuda.current_stream(dev).cuda_stream
        n_env, v_env, shared, exy, env = self.n_env, self.v_env, 1, self.env_xy, self.env_nv
        if env_xy is not None:
            if (env_xy.dim() != 4 or env_xy.shape[0] != B or env_xy.shape[3] != 2 or env_nv is None
                    or tuple(env_nv.shape) != (B, env_xy.shape[1]) or env_xy.dtype != torch.float64
                    or env_nv.dtype != torch.int32 or not env_xy.is_contiguous() or not env_nv.is_contiguous()
                    or env_xy.device != dev or env_nv.device != dev):
                raise ValueError("per-robot maps: env_xy [B,n_env,v_env,2] float64, env_nv [B,n_env] int32, contiguous, on the sensor's device")
            n_env, v_env, shared, exy, env = int(env_xy.shape[1]), int(env_xy.shape[2]), 0, env_xy, env_nv
        head = (self.device_index, B, self.resolution, n_env, v_env, shared, self.lidar_range, DBSCAN_EPS,
                DBSCAN_MIN_SAMPLES, self.n_obs_max, self.v_max, _ptr(state), _ptr(exy), _ptr(env), _ptr(self.table), _ptr(noise))
        if isinstance(schedule, str):
            if schedule != "auto":
                raise ValueError('schedule: a make_schedule(B) buffer, None or "auto"')
            schedule = None
            if want_ce and B > 2048:                         # beyond one round of waves (two per SIMD) the start order matters
                # one buffer per (batch size, stream): scans sharing a schedule must be ordered on one stream -- the order
                # kernel of one launch rewrites what the next one reads
                cache = self.__dict__.setdefault("_auto_sched", {})
                key = (B, stream)
                if key not in cache:
                    cache[key] = self.make_schedule(B)
                schedule = cache[key]
        if schedule is not None and (not want_ce or schedule.dtype != torch.int32 or schedule.device != dev or not schedule.is_contiguous()
                                     or schedule.numel() != int(self.lib.lipmpc_lidar_schedule_words(B))):
            raise ValueError("schedule: a buffer of make_schedule(B) for this B, with c_eta=True")
        if want_ce:
            rc = self.lib.lipmpc_lidar_c_eta_batch(*head, _ptr(out["c_eta"]), _ptr(out["n_inferred"]), _ptr(out["overflow"]),
                                                   _ptr(out.get("obs_xy")), _ptr(out.get("obs_nv")), _ptr(out.get("hits")),
                                                   _ptr(out.get("labels")), _ptr(schedule), C.c_void_p(stream))
            _lib.check(rc, "lipmpc_lidar_c_eta_batch")
        else:
            rc = self.lib.lipmpc_lidar_sense_batch(*head, _ptr(out["obs_xy"]), _ptr(out["obs_nv"]), _ptr(out["n_inferred"]),
                                                   _ptr(out["overflow"]), _ptr(out.get("hits")), _ptr(out.get("labels")),
                                                   C.c_void_p(stream))
            _lib.check(rc, "lipmpc_lidar_sense_batch")
        return out


    def sense_plan_step(self, solver, state, goal, first_foot, noise=None, delta=None, sen=None, out=None, schedule=None,
                        bounds=None):
        """One MPC step of the unknown-environment variant in one C call (lipmpc_sense_plan_step_batch): scan + constraint
        assembly, then ``solver``'s step against the assembled half-spaces.  ``solver``: a BatchedLipMpc whose
        n_obs_max / v_max are this sensor's.  Returns (sen, out) as ``sense(..., c_eta=True, rings=False)`` and
        ``plan_step_batch_c_eta`` would."""
        P = solver.params
        if P.n_obs_max != self.n_obs_max or P.v_max != self.v_max or solver.device != self.device:
            raise ValueError("solver and sensor must share n_obs_max, v_max and the device")
        B = solver._check_inputs(state, goal, first_foot, None, None, delta, need_obstacles=False)
        solver._check_optional(bounds, (B, 4), torch.float64, "bounds")
        if sen is None:
            sen = self.alloc_outputs(B, rings=False, c_eta=True)
        if out is None:
            out = solver.alloc_outputs(B)
        else:
            solver._check_outputs(out, B)
        for name, shape, dt in (("c_eta", (B, self.n_obs_max, 4), torch.float64), ("n_inferred", (B,), torch.int32), ("overflow", (B,), torch.int32)):
            t = sen.get(name)
            if t is None or tuple(t.shape) != shape or t.dtype != dt or t.device != self.device or not t.is_contiguous():
                raise ValueError(f"sen['{name}']: expected contiguous {dt} {shape} on {self.device}")
        if noise is not None and (tuple(noise.shape) != (B, self.resolution, 2) or noise.dtype != torch.float64
                                  or noise.device != self.device or not noise.is_contiguous()):
            raise ValueError(f"noise: expected contiguous float64 {(B, self.resolution, 2)} on {self.device}")
        if schedule is not None and (schedule.dtype != torch.int32 or schedule.device != self.device or not schedule.is_contiguous()
                                     or schedule.numel() != int(self.lib.lipmpc_lidar_schedule_words(B))):
            raise ValueError("schedule: a buffer of make_schedule(B) for this B")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.lipmpc_sense_plan_step_batch(
            solver._h, B, self.resolution, self.n_env, self.v_env, 1, self.lidar_range, DBSCAN_EPS, DBSCAN_MIN_SAMPLES,
            _ptr(state), _ptr(goal), _ptr(first_foot), _ptr(delta), _ptr(self.env_xy), _ptr(self.env_nv), _ptr(self.table),
            _ptr(noise), _ptr(sen["c_eta"]), _ptr(sen["n_inferred"]), _ptr(sen["overflow"]), _ptr(schedule),
            _ptr(out["U"]), _ptr(out["X"]), _ptr(out["theta"]), _ptr(out["omega"]), _ptr(out["obj"]), _ptr(out["status"]),
            _ptr(out["iters"]), _ptr(out["active"]), _ptr(out.get("working")), _ptr(out.get("diag")), _ptr(bounds), C.c_void_p(stream))
        _lib.check(rc, "lipmpc_sense_plan_step_batch")
        return sen, out


class HumanoidMPCUnknownEnvironment(HumanoidMPC):
    """The robot only perceives obstacles through its LiDAR (HumanoidMPCUnknownEnvironment.py:13-28): every sample the
    obstacle set is re-inferred on the GPU and handed to the step solver.  ``noise_seed`` seeds the readings' noise
    (the reference's is unseeded); ``noise_seed=None`` = noiseless readings."""

    def __init__(self, goal, obstacles, N_horizon=3, N_mpc_timesteps=100, sampling_time=1e-3, init_state=None,
                 start_with_right_foot: bool = True, verbosity: int = 1, lidar_range: float = 3.0,
                 lidar_resolution: int = 360, noise_seed: int | None = 0, **kw):
        self.lidar_range, self.lidar_resolution = lidar_range, lidar_resolution
        super().__init__(goal, obstacles, N_horizon, N_mpc_timesteps, sampling_time,
                         np.zeros(5) if init_state is None else init_state, start_with_right_foot, verbosity, **kw)
        # the reference scans `ch.points` (raw input order), HumanoidMPCUnknownEnvironment.py:46
        env = [np.asarray(o.points, float) if hasattr(o, "points") else np.asarray(o, float) for o in obstacles]
        self._env = env
        self._sensor = LidarSensor(env, lidar_range, lidar_resolution, device=self._device)
        self._big_sensor = None
        self._gen = None if noise_seed is None else torch.Generator(device=self._sensor.device).manual_seed(int(noise_seed))
        self.list_inferred_obstacles = []
        self.list_lidar_readings = []          # per scan: `resolution` entries, None or the noisy hit (x, y) -- the reference's
                                               # range_finder readings (HumanoidMPCUnknownEnvironment.py:66, HumanoidMpc.py:86)

    def _sense(self, x_k: float, y_k: float):
        """One scan at (x_k, y_k): (c_eta [n,4], rings) of the inferred obstacles, c / eta assembled in the scan's launch."""
        dev = self._sensor.device
        st = torch.tensor([[x_k, 0.0, y_k, 0.0, 0.0]], dtype=torch.float64, device=dev)
        noise = None
        if self._gen is not None:
            noise = NOISE_STD * torch.randn((1, self.lidar_resolution, 2), dtype=torch.float64, device=dev, generator=self._gen)
        out = self._sensor.sense(st, noise, c_eta=True, with_debug=True)
        torch.cuda.synchronize(dev)
        if int(out["overflow"][0]):
            # more clusters / longer hulls than the default slots: scan again (same noise) into the largest layout the
            # step solver takes; the reference constrains against every inferred obstacle (:55-64), so a scan that
            # still does not fit is an error, not a truncated obstacle list
            if self._big_sensor is None:
                self._big_sensor = LidarSensor(self._env, self.lidar_range, self.lidar_resolution, n_obs_max=50, v_max=32,
                                               device=self._device)
            out = self._big_sensor.sense(st, noise, c_eta=True, with_debug=True)
            torch.cuda.synchronize(dev)
            if int(out["overflow"][0]):
                raise RuntimeError("LiDAR scan inferred more obstacles / hull vertices than the solver holds (50 x 32)")
        n = int(out["n_inferred"][0])
        nv = out["obs_nv"][0].cpu().numpy()
        xy = out["obs_xy"][0].cpu().numpy()
        rings = [xy[j, :nv[j]].copy() for j in range(n)]
        self.list_inferred_obstacles.append(rings)
        hits = out["hits"][0].cpu().numpy()
        self.list_lidar_readings.append([None if h[0] != h[0] else (float(h[0]), float(h[1])) for h in hits])
        return out["c_eta"][0, :n].cpu().numpy(), rings

    def _get_list_c_and_eta(self, x_k: float, y_k: float):
        """The hook the reference's variant overrides (HumanoidMPCUnknownEnvironment.py:30-68): scan, cluster, hulls,
        closest point and normal per hull -- one launch (lipmpc_lidar_c_eta_batch); the step is then solved against
        these half-spaces through lipmpc_plan_step_batch_c_eta."""
        ce, _ = self._sense(x_k, y_k)
        return [r[:2].reshape(2, 1) for r in ce], [r[2:].reshape(2, 1) for r in ce]

    def _get_obstacle_rings(self, x_k: float, y_k: float):
        """The inferred obstacles as rings (one scan), for callers that want the polygons."""
        return self._sense(x_k, y_k)[1]


class UnknownEnvFleet:
    """B robots walking through one map that they only see through their LiDAR: the closed loop of
    HumanoidMPCUnknownEnvironment (HumanoidMpc.py:380-459 with _get_list_c_and_eta from
    HumanoidMPCUnknownEnvironment.py:30-68) for a whole batch and without a host round trip per sample — scan, step
    solve and state advance are enqueued back to back; with ``use_graph`` one sample is captured in a HIP graph and
    replayed.  One MPC solve per sample (sampling_time = DELTA_T), the reference's stop rule (previous objective <
    0.05) and stop-on-failed-solve per robot."""

    def __init__(self, env_rings, N_horizon=3, lidar_range=3.0, resolution=360, n_obs_max=12, v_max=32,
                 exact=False, interior_tol=1e-6, device=None):
        from .solver import BatchedLipMpc, LipMpcParams, FLAG_INTERIOR
        self.sensor = LidarSensor(env_rings, lidar_range, resolution, n_obs_max, v_max, device)
        self.solver = BatchedLipMpc(LipMpcParams(N=N_horizon, n_obs_max=n_obs_max, v_max=v_max,
                                                 flags=0 if exact else FLAG_INTERIOR, tol_interior=interior_tol),
                                    self.sensor.device_index)
        self.device = self.sensor.device

    def _plan_for(self, B, k_max, noise_mode, have_delta, stop_obj, use_graph):
        """Buffers (and, once captured, the HIP graph of one sample) of a run shape; kept across ``run`` calls, so a
        second run of the same shape replays the graph it already has."""
        key = (B, k_max, noise_mode, have_delta, float(stop_obj), bool(use_graph))
        pl = getattr(self, "_plan", None)
        if pl is not None and pl["key"] == key:
            return pl
        dev, sn, sv = self.device, self.sensor, self.solver
        f64 = dict(dtype=torch.float64, device=dev)
        fl = dict(state=torch.zeros((B, 5), **f64), first_foot=torch.ones((B,), dtype=torch.int8, device=dev),
                  walking=torch.ones((B,), dtype=torch.int8, device=dev), last_obj=torch.zeros((B,), **f64),
                  n_steps=torch.zeros((B,), dtype=torch.int32, device=dev),
                  last_status=torch.zeros((B,), dtype=torch.int32, device=dev),
                  n_overflow=torch.zeros((B,), dtype=torch.int32, device=dev),
                  sample=torch.zeros((1,), dtype=torch.int32, device=dev),
                  X_pred=torch.zeros((B, k_max + 1, 5), **f64), U_pred=torch.zeros((B, k_max, 3), **f64))
        pl = dict(key=key, fl=fl, goal=torch.zeros((B, 2), **f64), delta=torch.zeros((B,), **f64) if have_delta else None,
                  sen=sn.alloc_outputs(B, rings=False, c_eta=True),      # hulls stay in the scan kernel: only (c, eta) rows reach HBM
                  out=sv.alloc_outputs(B), nbuf=None if noise_mode == "none" else torch.zeros((B, sn.resolution, 2), **f64),
                  gen=torch.Generator(device=dev) if noise_mode == "seeded" else None, graph=None,
                  sched=sn.make_schedule(B))                 # order buffer: every scan ranks its robots before it starts them
        self._plan = pl
        return pl

    def run(self, state0, goal, first_foot, k_max, noise="seeded", noise_seed=0, delta=None, stop_obj=0.05,
            use_graph=True):
        """state0 [B,5], goal [B,2], first_foot [B] int8.  noise: "seeded" (N(0, 0.01) per reading from a generator
        seeded with noise_seed), None (noiseless) or a tensor [k_max,B,resolution,2].  Returns dict(X_pred
        [B,k_max+1,5], U_pred [B,k_max,3], n_steps [B] solved samples, last_status [B] (STATUS_SENSOR_OVERFLOW = 5: the
        robot was stopped because a scan's clusters did not fit the obstacle slots), overflow [B] number of such scans).
        One sample = noise draw (seeded mode), scan + constraint assembly, step solve, fleet update; with ``use_graph``
        it is captured once per run shape in a HIP graph (kept by the object) and replayed k_max times back to back.
        The returned tensors are the object's buffers: the next ``run`` of the same shape overwrites them."""
        dev, sv, sn = self.device, self.solver, self.sensor
        B = state0.shape[0]
        mode = "none" if noise is None else ("seeded" if isinstance(noise, str) else "given")
        pl = self._plan_for(B, int(k_max), mode, delta is not None, stop_obj, use_graph)
        fl, sen, out, nbuf, gen = pl["fl"], pl["sen"], pl["out"], pl["nbuf"], pl["gen"]
        pl["goal"].copy_(goal)
        if delta is not None:
            pl["delta"].copy_(delta)

        def reset():
            fl["state"].copy_(state0); fl["first_foot"].copy_(first_foot)
            fl["walking"].fill_(1); fl["last_obj"].fill_(float("inf"))
            for n in ("n_steps", "last_status", "n_overflow", "sample"):
                fl[n].zero_()
            fl["X_pred"].zero_(); fl["U_pred"].zero_(); fl["X_pred"][:, 0] = state0
            if gen is not None:
                gen.manual_seed(int(noise_seed))

        def sample():
            # HumanoidMpc.py:387/:417 sense + solve; :392 stop rule, :419-429 failed solve ends the run, :432-447 advance
            # and the trajectory row: one bookkeeping launch (lipmpc_fleet_update_batch)
            if gen is not None:
                nbuf.normal_(0.0, NOISE_STD, generator=gen)
            sn.sense_plan_step(sv, fl["state"], pl["goal"], fl["first_foot"], nbuf, pl["delta"], sen=sen, out=out, schedule=pl["sched"])
            sv.fleet_update(fl, out, overflow=sen["overflow"], stop_obj=stop_obj)

        if use_graph and pl["graph"] is None:
            reset()
            if mode == "given":
                nbuf.copy_(noise[0])
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                sample()                                     # warm-up outside capture (lazy initialisation)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            if gen is not None:
                graph.register_generator_state(gen)          # the draw is part of the graph: every replay advances the stream
            with torch.cuda.graph(graph):
                sample()
            pl["graph"] = graph
        reset()
        for k in range(k_max):
            if mode == "given":
                nbuf.copy_(noise[k])
            if use_graph:
                pl["graph"].replay()
            else:
                sample()
        return dict(X_pred=fl["X_pred"], U_pred=fl["U_pred"], n_steps=fl["n_steps"], last_status=fl["last_status"],
                    overflow=fl["n_overflow"])
