"""Generate golden vectors by IMPORTING the reference (build container only; the reference
never travels to the GPU box).  Run:

    PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs (committed, data only):
  geometry_golden.npz  rings (padded CCW vertex rings = hull.points[hull.vertices]), counts,
                       query points, and the reference's own outputs of
                       ObstaclesUtils.get_closest_point_and_normal_vector_from_obs (c, eta) and
                       ObstaclesUtils.is_point_inside_polygon (inside)   [ObstaclesUtils.py:50-109]
  fields_cfg2.npz      generate_obstacles fields, seeds 0..255, the BASELINE config-2 call
                       (obstacles.py:198-206): rings padded to 5 vertices + counts
  fields_cfg4.npz      same generator, 50 obstacles, box (0.5,15.5)^2, seeds 0..31
  scenario_circles.npz Scenario.CIRCLE_OBSTACLES rings (Scenario.py:202-210)
"""
import os

import numpy as np

from HumanoidNavigation.Utils import obstacles as ref_obstacles
from HumanoidNavigation.Utils.ObstaclesUtils import ObstaclesUtils
from HumanoidNavigation.report_simulations.Scenario import Scenario

HERE = os.path.dirname(os.path.abspath(__file__))


def ring_of(hull):
    return np.asarray(hull.points, float)[hull.vertices]


def pad_rings(hulls, vmax):
    rings = np.zeros((len(hulls), vmax, 2))
    nv = np.zeros(len(hulls), np.int32)
    for i, h in enumerate(hulls):
        r = ring_of(h)
        assert len(r) <= vmax
        rings[i, :len(r)] = r
        nv[i] = len(r)
    return rings, nv


def field(seed, n_obs, hi, goal):
    ref_obstacles.set_seed(seed)
    return ref_obstacles.generate_obstacles(start=(0, 0), goal=goal, num_obstacles=n_obs, num_points=5,
                                            x_range=(0.5, hi), y_range=(0.5, hi), delta=1)


def main():
    rng = np.random.default_rng(20251004)
    # ---- geometry goldens -------------------------------------------------------------
    hulls = []
    _, _, circ = Scenario.load_scenario(Scenario.CIRCLE_OBSTACLES, start=(0, 3), goal=(6, -3))
    hulls += circ
    _, _, paper = Scenario.load_scenario(Scenario.MAIN_PAPER, start=(0, 0), goal=(10, 10))
    hulls += paper
    for seed in range(12):
        hulls += field(seed, 10, 9.5, (10, 10))
    rings, nv = pad_rings(hulls, 24)
    pts, cs, etas, ins, which = [], [], [], [], []
    for i, h in enumerate(hulls):
        r = ring_of(h)
        ctr = r.mean(axis=0)
        ext = np.abs(r - ctr).max()
        qs = [ctr + rng.uniform(-3 * ext, 3 * ext, 2) for _ in range(24)]
        qs += [ctr + rng.uniform(-0.6 * ext, 0.6 * ext, 2) for _ in range(8)]      # mostly inside
        for a in range(len(r)):                                                      # near edges / vertices
            b = (a + 1) % len(r)
            t = rng.uniform(-0.2, 1.2)
            nrm = np.array([r[b][1] - r[a][1], -(r[b][0] - r[a][0])])
            nrm /= np.linalg.norm(nrm)
            qs.append(r[a] + t * (r[b] - r[a]) + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-9, -1) * nrm)
        for q in qs:
            c, eta = ObstaclesUtils.get_closest_point_and_normal_vector_from_obs(
                x=np.asarray(q, float), polygon=h, unitary_normal_vector=True)
            pts.append(q); cs.append(c.ravel()); etas.append(eta.ravel())
            ins.append(bool(ObstaclesUtils.is_point_inside_polygon(np.asarray(q, float), h)))
            which.append(i)
    np.savez_compressed(os.path.join(HERE, "geometry_golden.npz"), rings=rings, nv=nv,
                        pts=np.array(pts), c=np.array(cs), eta=np.array(etas),
                        inside=np.array(ins), which=np.array(which, np.int32))
    print("geometry:", len(pts), "queries on", len(hulls), "polygons; inside frac", np.mean(ins))
    # ---- benchmark-distribution obstacle fields ----------------------------------------
    R, V = [], []
    for seed in range(256):
        f = field(seed, 10, 9.5, (10, 10))
        assert len(f) == 10
        r, n = pad_rings(f, 5)
        R.append(r); V.append(n)
    np.savez_compressed(os.path.join(HERE, "fields_cfg2.npz"), rings=np.array(R), nv=np.array(V))
    R, V = [], []
    for seed in range(32):
        f = field(seed, 50, 15.5, (16, 16))
        r, n = pad_rings(f, 5)
        rr = np.zeros((50, 5, 2)); nn = np.zeros(50, np.int32)
        rr[:len(f)] = r; nn[:len(f)] = n
        R.append(rr); V.append(nn)
    np.savez_compressed(os.path.join(HERE, "fields_cfg4.npz"), rings=np.array(R), nv=np.array(V))
    print("cfg4 obstacle counts:", [int((v > 0).sum()) for v in V])
    rc, nc = pad_rings(circ, 24)
    np.savez_compressed(os.path.join(HERE, "scenario_circles.npz"), rings=rc, nv=nc)


if __name__ == "__main__":
    main()
