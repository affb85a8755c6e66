"""Golden vectors for the LiDAR front end (BASELINE config 5), produced by IMPORTING the reference (build container):

    PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_lidar_golden.py

For seeded CROWDED-style fields (Scenario.py:54-69, 20 obstacles in (-1,6)^2) and several robot positions:
the reference's own compute_lidar_readings (noiseless hits), range_finder with numpy's global generator seeded
(noisy readings -> the noise it drew is recovered by subtraction), the clusters scikit-learn's DBSCAN formed and
the convex rings build_local_obstacles returned.  Output: lidar_golden.npz (data only).
"""
import os

import numpy as np

from HumanoidNavigation.RangeFinder import range_finder_wth_polygons_dbscan as rf
from HumanoidNavigation.Utils import obstacles as ro
from HumanoidNavigation.report_simulations.Scenario import Scenario

HERE = os.path.dirname(os.path.abspath(__file__))
VMAX_ENV, N_ENV, R, MAX_INF, VMAX_INF = 5, 20, 360, 12, 40


def main():
    rng = np.random.default_rng(7)
    cases = []
    for seed in range(6):
        ro.set_seed(seed)
        _, _, obs = Scenario.load_scenario(Scenario.CROWDED, start=(0, 0), goal=(5, 5), num_max_obstacles=N_ENV,
                                           range_x=(-1, 6), range_y=(-1, 6))
        rings = [np.asarray(o.points, float) for o in obs]
        for _ in range(5):
            pos = rng.uniform(-0.5, 5.5, 2)
            if any(ro.is_point_inside_polygon(pos, list(map(tuple, r))) for r in rings):
                continue
            for lidar_range in (1.5, 3.0):
                clean = rf.compute_lidar_readings(pos, rings, lidar_range=lidar_range, resolution=R)
                np.random.seed(1000 * seed + len(cases))
                noisy, clusters, local = rf.range_finder(pos, rings, lidar_range=lidar_range, resolution=R)
                cases.append((pos, rings, lidar_range, clean, noisy, clusters, local))
    C = len(cases)
    env = np.zeros((C, N_ENV, VMAX_ENV, 2)); env_nv = np.zeros((C, N_ENV), np.int32)
    pos = np.zeros((C, 2)); rng_ = np.zeros(C)
    clean = np.zeros((C, R, 2)); valid = np.zeros((C, R), bool); noise = np.zeros((C, R, 2))
    labels = np.full((C, R), -2, np.int32)                      # -2 = no reading, -1 = DBSCAN noise
    inf_xy = np.zeros((C, MAX_INF, VMAX_INF, 2)); inf_nv = np.zeros((C, MAX_INF), np.int32)
    for i, (p, rings, lr, cl, no, clusters, local) in enumerate(cases):
        pos[i] = p; rng_[i] = lr
        for j, r in enumerate(rings):
            env[i, j, :len(r)] = r; env_nv[i, j] = len(r)
        for k in range(R):
            if cl[k] is not None:
                valid[i, k] = True; clean[i, k] = cl[k]; noise[i, k] = np.array(no[k]) - np.array(cl[k])
        # cluster membership per reading, from the reference's cluster arrays (points are unique with noise)
        pts = {tuple(np.round(np.array(no[k]), 14)): k for k in range(R) if no[k] is not None}
        labels[i][valid[i]] = -1
        for lab, cpts in enumerate(clusters):
            for q in cpts:
                labels[i][pts[tuple(np.round(q, 14))]] = lab
        assert len(local) <= MAX_INF
        for j, ring in enumerate(local):
            ring = ring[:-1]                                    # build_local_obstacles appends the first vertex again
            assert len(ring) <= VMAX_INF
            inf_xy[i, j, :len(ring)] = ring; inf_nv[i, j] = len(ring)
    np.savez_compressed(os.path.join(HERE, "lidar_golden.npz"), pos=pos, lidar_range=rng_, env=env, env_nv=env_nv,
                        clean=clean, valid=valid, noise=noise, labels=labels, inf_xy=inf_xy, inf_nv=inf_nv)
    print("cases", C, "valid readings per case", valid.sum(1).tolist())
    print("clusters per case", [int(l.max()) + 1 for l in labels], "inferred obstacles", (inf_nv > 0).sum(1).tolist(),
          "max hull vertices", inf_nv.max())


if __name__ == "__main__":
    main()
