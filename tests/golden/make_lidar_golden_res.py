"""More golden vectors for the LiDAR front end, at OTHER resolutions and ranges than lidar_golden.npz (360 rays, 1.5 / 3 m),
produced by IMPORTING the reference (build container):

    PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_lidar_golden_res.py

Same recipe as make_lidar_golden.py (CROWDED-style fields, the reference's compute_lidar_readings / range_finder with numpy's
global generator seeded so that the noise it drew can be recovered by subtraction, scikit-learn's clusters, Qhull's rings),
for resolution in {90, 180, 270} and lidar_range in {1.0, 2.0}.  Output: lidar_golden_res.npz (data only; arrays are padded
to 360 rays, `res` holds each case's resolution).
"""
import os

import numpy as np

from HumanoidNavigation.RangeFinder import range_finder_wth_polygons_dbscan as rf
from HumanoidNavigation.Utils import obstacles as ro
from HumanoidNavigation.report_simulations.Scenario import Scenario

HERE = os.path.dirname(os.path.abspath(__file__))
VMAX_ENV, N_ENV, RPAD, MAX_INF, VMAX_INF = 5, 20, 360, 12, 40


def main():
    rng = np.random.default_rng(11)
    cases = []
    for seed in range(3):
        ro.set_seed(20 + seed)
        _, _, obs = Scenario.load_scenario(Scenario.CROWDED, start=(0, 0), goal=(5, 5), num_max_obstacles=N_ENV,
                                           range_x=(-1, 6), range_y=(-1, 6))
        rings = [np.asarray(o.points, float) for o in obs]
        for _ in range(4):
            pos = rng.uniform(-0.5, 5.5, 2)
            if any(ro.is_point_inside_polygon(pos, list(map(tuple, r))) for r in rings):
                continue
            for R, lidar_range in ((90, 2.0), (180, 1.0), (180, 2.0), (270, 1.0)):
                clean = rf.compute_lidar_readings(pos, rings, lidar_range=lidar_range, resolution=R)
                np.random.seed(5000 + 100 * seed + len(cases))
                noisy, clusters, local = rf.range_finder(pos, rings, lidar_range=lidar_range, resolution=R)
                cases.append((pos, rings, lidar_range, R, clean, noisy, clusters, local))
    C = len(cases)
    env = np.zeros((C, N_ENV, VMAX_ENV, 2)); env_nv = np.zeros((C, N_ENV), np.int32)
    pos = np.zeros((C, 2)); rng_ = np.zeros(C); res = np.zeros(C, np.int32)
    clean = np.zeros((C, RPAD, 2)); valid = np.zeros((C, RPAD), bool); noise = np.zeros((C, RPAD, 2))
    labels = np.full((C, RPAD), -2, np.int32)                   # -2 = no reading, -1 = DBSCAN noise
    inf_xy = np.zeros((C, MAX_INF, VMAX_INF, 2)); inf_nv = np.zeros((C, MAX_INF), np.int32)
    for i, (p, rings, lr, R, cl, no, clusters, local) in enumerate(cases):
        pos[i] = p; rng_[i] = lr; res[i] = R
        for j, r in enumerate(rings):
            env[i, j, :len(r)] = r; env_nv[i, j] = len(r)
        for k in range(R):
            if cl[k] is not None:
                valid[i, k] = True; clean[i, k] = cl[k]; noise[i, k] = np.array(no[k]) - np.array(cl[k])
        pts = {tuple(np.round(np.array(no[k]), 14)): k for k in range(R) if no[k] is not None}
        labels[i, :R][valid[i, :R]] = -1
        for lab, cpts in enumerate(clusters):
            for q in cpts:
                labels[i][pts[tuple(np.round(q, 14))]] = lab
        assert len(local) <= MAX_INF
        for j, ring in enumerate(local):
            ring = ring[:-1]                                    # build_local_obstacles appends the first vertex again
            assert len(ring) <= VMAX_INF
            inf_xy[i, j, :len(ring)] = ring; inf_nv[i, j] = len(ring)
    np.savez_compressed(os.path.join(HERE, "lidar_golden_res.npz"), pos=pos, lidar_range=rng_, res=res, env=env, env_nv=env_nv,
                        clean=clean, valid=valid, noise=noise, labels=labels, inf_xy=inf_xy, inf_nv=inf_nv)
    print("cases", C, "resolutions", res.tolist(), "valid readings per case", valid.sum(1).tolist())
    print("inferred obstacles", (inf_nv > 0).sum(1).tolist(), "max hull vertices", inf_nv.max())


if __name__ == "__main__":
    main()
