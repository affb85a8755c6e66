"""The rules by which the scan kernel clusters a scan as chains of consecutive readings (csrc/lipmpc_lidar_chains.inc), restated
in numpy and fuzzed against the oracle's DBSCAN (oracle/lidar_oracle.py::dbscan_labels, itself pinned to scikit-learn): whenever
the rules claim an answer -- they may always decline, the kernel then takes the general route -- it must be DBSCAN's labels.  The
GPU tests (tests/test_lidar.py::test_gpu_clustering_routes_against_oracle) check the kernel on real scans; this one checks the
ARGUMENT on adversarial point sequences no map would produce often: steps around eps, sparse piece ends, short pieces beside
long ones, pieces that come back near earlier ones, sequences closing on their first reading."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lidar_oracle as L  # noqa: E402

NO_ROOT = 1 << 30
SEGMAX = 8


def chain_labels(P, eps, ms):
    """Labels by the chain rules, or None where the rules decline.  Mirrors lipmpc_lidar_chains.inc step by step."""
    n = len(P)
    if n < 1:
        return None
    eps2 = eps * eps
    D2 = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)                  # dx*dx + dy*dy, as the kernel forms it
    nb = D2 <= eps2
    link = np.array([i + 1 < n and nb[i, i + 1] for i in range(n)])
    ends = np.nonzero(~link)[0]
    nseg = len(ends)
    if nseg > SEGMAX:
        return None
    starts = np.concatenate([[0], ends[:-1] + 1])
    lens = ends - starts + 1
    pc = np.searchsorted(ends, np.arange(n))                             # piece of every reading
    box = [(P[a:b + 1, 0].min(), P[a:b + 1, 0].max(), P[a:b + 1, 1].min(), P[a:b + 1, 1].max()) for a, b in zip(starts, ends)]
    epsx = eps * (1.0 + 1e-6) + 1e-9
    near = lambda A, B: B[0] <= A[1] + epsx and B[1] >= A[0] - epsx and B[2] <= A[3] + epsx and B[3] >= A[2] - epsx
    wrap_close = n >= 4 and nb[n - 1, 0]
    joined = wrap_close and nseg >= 2
    # cheap neighbour count, suspects, exact counts
    noncore = set()
    suspects = []
    for i in range(n):
        st, en = starts[pc[i]], ends[pc[i]]
        lenj = lens[pc[i]] + (lens[-1] if joined and pc[i] == 0 else 0) + (lens[0] if joined and pc[i] == nseg - 1 else 0)
        cnt = int(i > st) + int(i < en) + int(i + 2 < n and nb[i, i + 2]) + int(i >= 2 and nb[i, i - 2])
        cnt += int(wrap_close and i in (0, n - 1))
        if lenj >= ms and cnt < ms - 1:
            suspects.append(i)
    if len(suspects) > 8:
        return None
    for e in suspects:
        if nb[e].sum() < ms:
            st, en = starts[pc[e]], ends[pc[e]]
            if (e != st and e != en) or (joined and e in (0, n - 1)):
                return None
            noncore.add(e)
    lab = list(range(nseg))

    def relabel(p, q):
        lo, hi = min(lab[p], lab[q]), max(lab[p], lab[q])
        for k in range(nseg):
            if lab[k] == hi:
                lab[k] = lo
    for t in range(1, nseg):
        for s in range(t):
            if not near(box[s], box[t]) or (joined and s == 0 and t == nseg - 1):
                continue
            cores_only = lens[s] >= ms and lens[t] >= ms
            found = False
            for i in range(starts[s], ends[s] + 1):
                for j in range(starts[t], ends[t] + 1):
                    if cores_only and (i in noncore or j in noncore):
                        continue
                    found |= bool(nb[i, j])
            if found:
                if not cores_only:
                    return None
                relabel(s, t)
    if joined:
        relabel(0, nseg - 1)
    mlen = [sum(lens[k] for k in range(nseg) if lab[k] == L_) for L_ in range(nseg)]
    core = np.array([mlen[lab[pc[i]]] >= ms and i not in noncore for i in range(n)])
    croot = {}
    for t in range(nseg):
        if lab[t] != t or mlen[t] < ms:
            continue
        idx = [i for i in range(n) if core[i] and lab[pc[i]] == t]
        if not idx:
            return None
        croot[t] = min(idx)
    root = np.array([croot[lab[pc[i]]] if core[i] else NO_ROOT for i in range(n)])
    for e in noncore:                                                     # border points: smallest root among core neighbours
        cand = [root[j] for j in range(n) if core[j] and nb[e, j]]
        root[e] = min(cand) if cand else NO_ROOT
    order = sorted(set(r for i, r in enumerate(root) if r != NO_ROOT and core[i]))
    k_of = {r: k for k, r in enumerate(order)}
    return np.array([k_of.get(r, -1) if r != NO_ROOT else -1 for r in root])


def _sequence(rng, eps):
    """An adversarial reading sequence: a walk whose steps are mostly small, sometimes around eps, sometimes a jump, which now
    and then returns to the neighbourhood of an earlier reading or of its first one."""
    n = int(rng.integers(1, 70))
    P = np.zeros((n, 2))
    p = rng.uniform(-1, 1, 2)
    heading = rng.uniform(0, 2 * np.pi)
    for i in range(n):
        P[i] = p
        u = rng.uniform()
        if u < 0.62:
            step = rng.uniform(0.01, 0.12)
        elif u < 0.80:
            step = eps * rng.uniform(0.7, 1.3)                           # around eps: piece ends, sparse ends
        elif u < 0.90:
            step = rng.uniform(0.4, 1.5)                                 # a jump: a new piece
        else:                                                            # come back near an earlier reading
            j = int(rng.integers(0, i + 1))
            p = P[j] + rng.uniform(-1, 1, 2) * eps * rng.uniform(0.3, 1.4)
            heading = rng.uniform(0, 2 * np.pi)
            continue
        heading += rng.normal(0, 0.5)
        p = p + step * np.array([np.cos(heading), np.sin(heading)])
    if n >= 4 and rng.uniform() < 0.3:                                   # close on the first reading
        P[-1] = P[0] + rng.uniform(-1, 1, 2) * eps * rng.uniform(0.2, 1.1)
    return P


def _short_piece_beside_a_sparse_end(rng, eps):
    """A short piece whose reading lies within eps of the sparse END of a long piece (a reading with too few neighbours to be a
    core point) -- not a core point itself, that end still is a neighbour: it can make the short piece's reading one."""
    k = int(rng.integers(1, 4))                                           # readings of the short piece
    short = np.cumsum(np.vstack([[0.0, 0.0]] + [rng.uniform(0.05, 0.9) * eps * np.array([1.0, 0.0]) for _ in range(k - 1)]), axis=0)
    m = int(rng.integers(4, 30))
    steps = np.full(m, 0.05)
    steps[-int(rng.integers(1, 3)):] = eps * rng.uniform(0.55, 0.95)     # the last one or two steps are sparse
    xs = np.concatenate([[0.0], np.cumsum(steps)])
    long_ = np.stack([xs, np.zeros_like(xs)], 1)
    # put the long piece's END at a random offset within ~eps of the short piece's last reading, the piece running away from it
    ang = rng.uniform(0, 2 * np.pi)
    rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    long_ = (long_ - long_[-1]) @ rot.T + short[-1] + rng.uniform(-1, 1, 2) * eps * rng.uniform(0.2, 1.2)
    P = np.vstack([short, long_]) if rng.uniform() < 0.5 else np.vstack([long_[::-1], short[::-1]])
    return P + rng.normal(0, 0.003, P.shape)


def _two_sparse_ends(rng, eps):
    """Two long pieces whose sparse ends (readings that may not be core points) come within eps of each other: a pair of
    neighbours, but only a pair of CORE points makes the two pieces one cluster."""
    def piece(m):
        steps = np.full(m, 0.04)
        steps[-int(rng.integers(1, 3)):] = eps * rng.uniform(0.55, 0.98)
        xs = np.concatenate([[0.0], np.cumsum(steps)])
        return np.stack([xs, np.zeros_like(xs)], 1)
    A, B = piece(int(rng.integers(3, 25))), piece(int(rng.integers(3, 25)))
    ang = rng.uniform(0, 2 * np.pi)
    rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    B = (B - B[-1]) @ rot.T + A[-1] + rng.uniform(-1, 1, 2) * eps * rng.uniform(0.2, 1.3)
    P = [np.vstack([A, B[::-1]]), np.vstack([A[::-1], B]), np.vstack([B, A]), np.vstack([A, B])][int(rng.integers(0, 4))]
    return P + rng.normal(0, 0.003, P.shape)


@pytest.mark.parametrize("ms", [1, 2, 3, 4, 5])
def test_chain_rules_never_contradict_dbscan(ms):
    rng = np.random.default_rng(100 + ms)
    eps = 0.3
    claimed = declined = 0
    for _ in range(4000):
        u = rng.uniform()
        P = _sequence(rng, eps) if u < 0.6 else (_short_piece_beside_a_sparse_end(rng, eps) if u < 0.8 else _two_sparse_ends(rng, eps))
        got = chain_labels(P, eps, ms)
        if got is None:
            declined += 1
            continue
        claimed += 1
        want = L.dbscan_labels(P, eps, ms)
        assert np.array_equal(got, want), (ms, len(P), P.tolist(), got.tolist(), want.tolist())
    assert claimed > 200, (claimed, declined)                            # the rules do claim a good share of even these
