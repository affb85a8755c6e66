"""CPU oracle (numpy, float64) of the reference's unknown-environment front end — BASELINE config 5.

TEST INFRASTRUCTURE ONLY (same rules as lipmpc_oracle.py).  It restates, per MPC step,

  HumanoidMPCUnknownEnvironment._get_list_c_and_eta     MPC/HumanoidMPCVariants/HumanoidMPCUnknownEnvironment.py:30-68
    range_finder                                        RangeFinder/range_finder_wth_polygons_dbscan.py:157-180
      compute_lidar_readings                            ... :26-63   (rays x obstacles x edges, nearest hit within range)
        line_polygon_intersection / compute_intersection  Utils/obstacles.py:95-139
        get_closest_point                               ... :14-24
      Gaussian noise sigma = 0.01 on valid readings     ... :162-172 (np.random.normal, global UNSEEDED generator)
      retrieve_clusters: DBSCAN(eps=0.3, min_samples=3) ... :100-116 (scikit-learn)
      build_local_obstacles / create_convex_hull        ... :65-83, 119-126 (np.unique, rank test, Qhull)
    ConvexHull(obstacle) again + closest point / normal ... UnknownEnvironment.py:54-62

Differences that are forced, and how parity is still pinned:
* the reference draws its noise from numpy's global generator without seeding, so its runs are not reproducible;
  here the noise is an explicit input array [R,2] (zeros = noiseless).  tests/golden/make_lidar_golden.py seeds the
  global generator, calls the reference, and recovers the noise it drew as (noisy reading - noiseless reading).
* scikit-learn's DBSCAN is restated through its defining property instead of its traversal: core points (>= 3
  neighbours within eps, the point itself included, distance <= eps) fall into connected components; clusters are
  numbered by their smallest core index (the order in which sklearn's index-ordered outer loop discovers them); a
  border point gets the smallest label among the clusters owning a core neighbour (the first cluster whose expansion
  reaches it).  Pinned against sklearn's labels on the reference's own readings.
* Qhull is restated by Andrew's monotone chain (strict turns): the same CCW ring of extreme points up to rotation;
  closest point / normal do not depend on where the ring starts.
"""
from __future__ import annotations

import math

import numpy as np

NOISE_STD = 0.01
DBSCAN_EPS = 0.3
DBSCAN_MIN_SAMPLES = 3


def ray_table(resolution=360):
    """unit directions of the rays, angle_i = i * (2 pi / resolution) (range_finder_wth_polygons_dbscan.py:28-36),
    through math.cos / math.sin exactly as the reference; shipped to the GPU as an input table."""
    step = 2 * math.pi / resolution
    return np.array([[math.cos(i * step), math.sin(i * step)] for i in range(resolution)])


def _segment_hit(a1, b1, a2, b2):
    """compute_intersection (Utils/obstacles.py:107-123): point of p1p2 ∩ q1q2 or None."""
    denom = (b2[1] - a2[1]) * (b1[0] - a1[0]) - (b2[0] - a2[0]) * (b1[1] - a1[1])
    if denom == 0:
        return None
    ua = ((b2[0] - a2[0]) * (a1[1] - a2[1]) - (b2[1] - a2[1]) * (a1[0] - a2[0])) / denom
    ub = ((b1[0] - a1[0]) * (a1[1] - a2[1]) - (b1[1] - a1[1]) * (a1[0] - a2[0])) / denom
    if 0 <= ua <= 1 and 0 <= ub <= 1:
        return (a1[0] + ua * (b1[0] - a1[0]), a1[1] + ua * (b1[1] - a1[1]))
    return None


def lidar_hits(position, rings, lidar_range, table):
    """compute_lidar_readings (:26-63): per ray the nearest intersection strictly closer than lidar_range, obstacles in
    list order, edges in ring order (ring[i] -> ring[i+1], closing edge last); ties keep the earlier one."""
    x, y = float(position[0]), float(position[1])
    R = len(table)
    hits = np.zeros((R, 2))
    valid = np.zeros(R, bool)
    for i in range(R):
        end = (x + lidar_range * table[i, 0], y + lidar_range * table[i, 1])
        best, best_d = None, lidar_range
        for ring in rings:
            close, close_d = None, lidar_range
            n = len(ring)
            for e in range(n):
                p = _segment_hit((x, y), end, ring[e], ring[(e + 1) % n])
                if p is None:
                    continue
                dist = math.sqrt((p[0] - x) * (p[0] - x) + (p[1] - y) * (p[1] - y))   # np.linalg.norm of a 2-vector
                if dist < close_d:
                    close, close_d = p, dist
            if close is not None and close_d <= lidar_range and close_d < best_d:
                best, best_d = close, close_d
        if best is not None:
            hits[i] = best
            valid[i] = True
    return hits, valid


def dbscan_labels(points, eps=DBSCAN_EPS, min_samples=DBSCAN_MIN_SAMPLES):
    """scikit-learn DBSCAN labels (-1 = noise) from the order-free characterisation in the module docstring."""
    n = len(points)
    if n == 0:
        return np.zeros(0, int)
    d2 = ((points[:, None, :] - points[None, :, :]) ** 2).sum(-1)
    nb = d2 <= eps * eps
    core = nb.sum(1) >= min_samples
    comp = np.where(core, np.arange(n), n)            # component id = smallest core index, by min-propagation
    while True:
        new = comp.copy()
        for i in np.where(core)[0]:
            new[i] = min(comp[i], comp[nb[i] & core].min())
        if np.array_equal(new, comp):
            break
        comp = new
    roots = sorted(set(comp[core].tolist()))
    label_of = {r: k for k, r in enumerate(roots)}
    labels = np.full(n, -1)
    for i in range(n):
        if core[i]:
            labels[i] = label_of[comp[i]]
        else:
            cn = nb[i] & core
            if cn.any():
                labels[i] = min(label_of[comp[j]] for j in np.where(cn)[0])
    return labels


def hull_ring(points):
    """create_convex_hull (:65-83): unique rows, < 3 points or collinear -> None, else the CCW ring of extreme points."""
    pts = np.unique(np.asarray(points, float), axis=0)
    if len(pts) < 3 or np.linalg.matrix_rank(pts - pts[0]) < 2:
        return None
    p = [tuple(r) for r in pts]                      # np.unique sorted them lexicographically

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


def range_finder(position, rings, lidar_range, noise=None, table=None):
    """range_finder (:157-180): returns (readings [R,2], valid [R], labels of the valid readings, inferred rings)."""
    table = ray_table() if table is None else table
    hits, valid = lidar_hits(position, rings, lidar_range, table)
    if noise is not None:
        hits = hits + np.where(valid[:, None], noise, 0.0)
    pts = hits[valid]
    labels = dbscan_labels(pts)
    out = []
    for k in range(labels.max() + 1 if len(labels) else 0):
        ring = hull_ring(pts[labels == k])
        if ring is not None:
            out.append(ring)
    return hits, valid, labels, out
