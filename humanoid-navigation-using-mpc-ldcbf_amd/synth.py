"""Host-side synthesis of benchmark inputs (not on the hot path).

``synthetic_fields`` restates the distribution of the reference's obstacle generator
(HumanoidNavigation/Utils/obstacles.py:167-206, the BASELINE config-2 call
``generate_obstacles(start, goal, num_obstacles, num_points=5, x_range, y_range, delta=1)``):
centre uniform in the box, 5 points uniform in the unit square around it, convex hull (3-5
vertices, CCW); a candidate is rejected when it contains start or goal, intersects an accepted
polygon, or its centre is closer than ``delta`` to one; at most 500 attempts.  It uses numpy's
generator, not CPython's ``random`` stream, so fields differ from the reference's seed-for-seed
(tests use committed fields produced by the reference itself, tests/golden/fields_cfg*.npz).

``walk_states`` produces reachable walking states by running the solver's own closed loop on the
GPU for a per-problem random number of steps from rest.
"""
from __future__ import annotations

import numpy as np
import torch


def _hull(pts):
    p = pts[np.lexsort((pts[:, 1], pts[:, 0]))]

    def half(seq):
        out = []
        for q in seq:
            while len(out) >= 2 and ((out[-1][0] - out[-2][0]) * (q[1] - out[-2][1])
                                     - (out[-1][1] - out[-2][1]) * (q[0] - out[-2][0])) <= 0:
                out.pop()
            out.append(q)
        return out
    lo, up = half(p), half(p[::-1])
    return np.array(lo[:-1] + up[:-1])


def _inside(pt, poly):
    a, b = poly, np.roll(poly, -1, axis=0)
    cr = (b[:, 0] - a[:, 0]) * (pt[1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (pt[0] - a[:, 0])
    return bool(np.all(cr >= 0))


def _dist_point_poly(pt, poly):
    a, b = poly, np.roll(poly, -1, axis=0)
    e = b - a
    t = np.clip(((pt - a) * e).sum(1) / (e * e).sum(1), 0.0, 1.0)
    c = a + t[:, None] * e
    return float(np.sqrt(((c - pt) ** 2).sum(1)).min())


def _sat_intersect(p, q):
    for poly in (p, q):
        e = np.roll(poly, -1, axis=0) - poly
        nrm = np.stack([e[:, 1], -e[:, 0]], axis=1)
        a, b = p @ nrm.T, q @ nrm.T
        if np.any(a.max(0) < b.min(0)) or np.any(b.max(0) < a.min(0)):
            return False
    return True


def synthetic_fields(B, n_obs, lo, hi, start, goal, seed, delta=1.0, v_max=5):
    rng = np.random.default_rng(seed)
    xy = np.zeros((B, n_obs, v_max, 2))
    nv = np.zeros((B, n_obs), np.int32)
    start, goal = np.asarray(start, float), np.asarray(goal, float)
    for b in range(B):
        polys, ctrs = [], []
        attempts = 0
        while len(polys) < n_obs and attempts < 500:
            attempts += 1
            c = rng.uniform(lo, hi, 2)
            poly = _hull(c + rng.uniform(-0.5, 0.5, (5, 2)))
            if len(poly) < 3 or _inside(start, poly) or _inside(goal, poly):
                continue
            ok = True
            for pc, pp in zip(ctrs, polys):
                d2 = (pc[0] - c[0]) ** 2 + (pc[1] - c[1]) ** 2
                if d2 >= 2.93:            # > (1 + sqrt(.5))^2: neither rule can trigger
                    continue
                if _dist_point_poly(c, pp) < delta or _sat_intersect(poly, pp):
                    ok = False
                    break
            if ok:
                polys.append(poly)
                ctrs.append(c)
        for j, poly in enumerate(polys):
            xy[b, j, : len(poly)] = poly
            nv[b, j] = len(poly)
    return xy, nv


def walk_states(solver_interior, obs_xy, obs_nv, goal, max_steps, seed, delta=None):
    """Closed-loop warm-up on the device: every robot starts at rest in the origin, right foot
    first, and walks w_b ~ U{0..max_steps} MPC steps; its benchmark state is the state it is in at
    step w_b.  A robot whose loop has already stopped (the reference breaks out of run_simulation
    on a failed solve, HumanoidMpc.py:419-429) contributes the last state from which its step was
    still solvable, i.e. the batch holds live robots only.  Returns (state [B,5], first_foot [B]
    int8) device tensors."""
    dev = solver_interior.device
    B = obs_xy.shape[0]
    gen = torch.Generator(device="cpu").manual_seed(seed)
    w = torch.randint(0, max_steps + 1, (B,), generator=gen).to(dev)
    state = torch.zeros((B, 5), dtype=torch.float64, device=dev)
    foot = torch.ones((B,), dtype=torch.int8, device=dev)
    keep_state, keep_foot = state.clone(), foot.clone()
    out = solver_interior.alloc_outputs(B)
    for k in range(max_steps + 1):
        solver_interior.plan_step_batch(state, goal, foot, obs_xy, obs_nv, delta, out=out)
        ok = (out["status"] == 0) | (out["status"] == 4)
        take = ok & (w >= k)                 # latest solvable state not beyond the robot's draw
        keep_state[take] = state[take]
        keep_foot[take] = foot[take]
        solver_interior.advance(state, foot, out)
    return keep_state.contiguous(), keep_foot.contiguous()
