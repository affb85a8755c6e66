"""Shared test helpers (CPU side)."""
import numpy as np

import lipmpc_oracle as O


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
