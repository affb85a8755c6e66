"""Everything the reference's committed result figures pin about its CasADi/IPOPT closed loop, as data:

    PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_pdf_pins.py

(build container only: reads /root/reference/Assets/ReportResults and imports the reference's Scenario module;
HumanoidMpc itself cannot be imported -- casadi is absent -- so no trajectory here comes from running the reference.)

Outputs (data only):
  pdf_series.npz      every polyline of Assets/ReportResults/<run>/evolutions*/evolution_<i>.pdf in data units
                      (extract_pdf_series.extract: time [s] = k * 0.4 against the plotted signal).  matplotlib's path
                      simplification drops nearly collinear vertices of long runs, so series are compared by their
                      time stamps, not by index.
  pdf_scenarios.npz   per run: obstacle rings (Scenario.load_scenario / the literal hulls of the scenario scripts),
                      init state, goal, horizon, and for the RRT* runs the sub-goal list recovered from rrt_res.pdf
                      (the red width-2 path of rrtplanner.plot_path, in occupancy-grid cells, mapped back to world
                      coordinates with the affine map of HumanoidMPCWithRRT._build_occupancy_grid, :44-64).
Runs: Simulation1 (simulation_1.py:33-50, BASE seed 7), Simulation1Circles (:85-102), Simulation1CirclesDelta (:146-160),
SimulationRRT-NoRRT (simulation_rrt.py:17-45), SimulationRRT (:67-84), SimulationMaze1 / SimulationMaze2
(simulation_maze.py:14-60 with MAZE_1 -> (7.5, 7.5) and MAZE_2 -> (0.5, 7.5)), Simulation4UnkEnv (simulation_1.py:195-232,
the unknown-environment run as committed: its map and start are reproducible, its sensor noise -- unseeded,
range_finder_wth_polygons_dbscan.py:162-172 -- is not, so it pins the closed loop to the centimetres sigma = 0.01 m of
noise leaves).  Simulation1..3UnkEnv start at theta = pi/4: produced by variants of the script that are not committed.
"""
import math
import os
import re
import zlib

import numpy as np
from scipy.spatial import ConvexHull

import extract_pdf_series as E
from HumanoidNavigation.report_simulations.Scenario import Scenario

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = "/root/reference/Assets/ReportResults"


def ring_of(h):
    return np.asarray(h.points, float)[h.vertices]


def pad(hulls, vmax=24):
    rings = np.zeros((len(hulls), vmax, 2)); nv = np.zeros(len(hulls), np.int32)
    for i, h in enumerate(hulls):
        r = ring_of(h); rings[i, :len(r)] = r; nv[i] = len(r)
    return rings, nv


def rrt_subgoals(path, hulls, goal, width=250):
    d = open(path, "rb").read()
    page = b""
    for m in re.finditer(rb"stream\r?\n(.*?)endstream", d, re.S):
        try:
            t = zlib.decompress(m.group(1))
        except zlib.error:
            continue
        if b" TJ" in t and len(t) > len(page):
            page = t
    txt = page.decode("latin1")
    xt, yt = E._ticks(txt)
    kx, bx, ex = E._calibrate(xt)
    ky, by, ey = E._calibrate(yt)
    ms = list(re.finditer(r"2 w 1 0\s+0 RG", txt))
    assert ms
    i = ms[-1].start()
    segs = re.findall(rf"({E.NUM}) ({E.NUM}) m\n({E.NUM}) ({E.NUM}) l\n\nS", txt[i:])
    ends = np.array([[float(s[2]), float(s[3])] for s in segs])
    starts = np.array([[float(s[0]), float(s[1])] for s in segs])
    assert np.allclose(starts[1:], ends[:-1])
    cells = np.stack([kx * ends[:, 0] + bx, ky * ends[:, 1] + by], axis=1)
    assert np.max(np.abs(cells - np.round(cells))) < 0.02, cells
    cells = np.round(cells)
    # HumanoidMPCWithRRT._build_occupancy_grid:32-64
    vs = np.vstack([ring_of(h) for h in hulls])
    min_x = min(0, goal[0], vs[:, 0].min()) - 3; max_x = max(0, goal[0], vs[:, 0].max()) + 3
    min_y = min(0, goal[1], vs[:, 1].min()) - 3; max_y = max(0, goal[1], vs[:, 1].max()) + 3
    height = math.ceil(width * ((max_y - min_y) / (max_x - min_x)))
    sub = np.stack([min_x + cells[:, 0] * (max_x - min_x) / width, min_y + cells[:, 1] * (max_y - min_y) / height], axis=1)
    first = np.array([kx * starts[0, 0] + bx, ky * starts[0, 1] + by])
    start_cell = np.array([round((0 - min_x) / (max_x - min_x) * width), round((0 - min_y) / (max_y - min_y) * height)])
    assert np.max(np.abs(first - start_cell)) < 0.02, (first, start_cell)      # the path starts at cell(0, 0)
    goal_cell = np.array([round((goal[0] - min_x) / (max_x - min_x) * width), round((goal[1] - min_y) / (max_y - min_y) * height)])
    assert np.max(np.abs(cells[-1] - goal_cell)) < 0.02, (cells[-1], goal_cell)
    return sub


def main():
    series = {}
    runs = [("Simulation1", "evolutions"), ("Simulation1Circles", "evolutions"), ("Simulation1CirclesDelta", "evolutions"),
            ("SimulationRRT-NoRRT", "evolutions"), ("SimulationRRT", "evolutions"), ("SimulationMaze1", "evolutions.pdf"),
            ("SimulationMaze2", "evolutions"), ("Simulation4UnkEnv", "evolutions")]
    for run, sub in runs:
        for i in range(6):
            p = f"{ROOT}/{run}/{sub}/evolution_{i}.pdf"
            if not os.path.exists(p):
                continue
            ser, err = E.extract(p)
            for j, s in enumerate(ser):
                series[f"{run}/ev{i}/s{j}"] = s
            print(run, i, [s.shape for s in ser], "tick fit err", err)
    np.savez_compressed(os.path.join(HERE, "pdf_series.npz"), **series)

    sc = {}

    def put(run, hulls, init, goal, N, delta=0.0, subgoals=None):
        r, n = pad(hulls)
        sc[run + "/rings"] = r; sc[run + "/nv"] = n
        sc[run + "/init"] = np.asarray(init, float); sc[run + "/goal"] = np.asarray(goal, float)
        sc[run + "/N"] = np.int32(N); sc[run + "/delta"] = np.float64(delta)
        if subgoals is not None:
            sc[run + "/subgoals"] = subgoals
            print(run, "sub-goals", subgoals.tolist())

    _, _, base = Scenario.load_scenario(Scenario.BASE, start=(0, 0), goal=(5, 5), seed=7)
    put("Simulation1", base, (0, 0, 0, 0, 0), (5, 5), 3)
    _, _, circ = Scenario.load_scenario(Scenario.CIRCLE_OBSTACLES, start=(0, 3), goal=(6, -3))
    put("Simulation1Circles", circ, (0, 0, 3, 0, 0), (6, -3), 3)
    put("Simulation1CirclesDelta", circ, (0, 0, 3, 0, 0), (6, -3), 3, delta=0.3)
    wall = [ConvexHull(np.array([[2, -3], [2, 3], [3, -3], [3, 3]]))]
    put("SimulationRRT-NoRRT", wall, (0, 0, 0, 0, 0), (5, 0), 3)
    put("SimulationRRT", wall, (0, 0, 0, 0, 0), (5, 0), 3, subgoals=rrt_subgoals(f"{ROOT}/SimulationRRT/rrt_res.pdf", wall, (5, 0)))
    _, g1, m1 = Scenario.load_scenario(Scenario.MAZE_1, (0.5, 0.5), (7.5, 7.5), 20, range_x=(-1, 6), range_y=(-1, 6))
    put("SimulationMaze1", m1, (0, 0, 0, 0, 0), g1, 3, subgoals=rrt_subgoals(f"{ROOT}/SimulationMaze1/rrt_res.pdf", m1, g1))
    _, g2, m2 = Scenario.load_scenario(Scenario.MAZE_2, (0.5, 0.5), (0.5, 7.5), 20, range_x=(-1, 6), range_y=(-1, 6))
    put("SimulationMaze2", m2, (0, 0, 0, 0, 0), g2, 3, subgoals=rrt_subgoals(f"{ROOT}/SimulationMaze2/rrt_res.pdf", m2, g2))
    # Simulation4UnkEnv = run_simulation_unk_env as committed (simulation_1.py:195-232): CROWDED map of 20 obstacles in
    # (-1, 6)^2 under seed 10, start (0, 0, pi/2), goal (4, 3.5), N = 3, lidar_range 1.5.  The sensor noise was unseeded
    # (range_finder_wth_polygons_dbscan.py:162-172), so the figure pins the closed loop only as far as sigma = 0.01 m of
    # sensor noise lets it: the map is stored as the raw `ch.points` arrays the reference scans (:46), in list order.
    from HumanoidNavigation.Utils.ObstaclesUtils import ObstaclesUtils
    from HumanoidNavigation.Utils.obstacles import set_seed
    ObstaclesUtils.set_random_seed(10); set_seed(10)
    _, g4, crowd = Scenario.load_scenario(Scenario.CROWDED, (0, 0), (4, 3.5), 20, range_x=(-1, 6), range_y=(-1, 6))
    put("Simulation4UnkEnv", crowd, (0, 0, 0, 0, math.pi / 2), g4, 3)
    vmax = max(len(h.points) for h in crowd)
    pts = np.zeros((len(crowd), vmax, 2)); npt = np.zeros(len(crowd), np.int32)
    for i, h in enumerate(crowd):
        pts[i, :len(h.points)] = np.asarray(h.points, float); npt[i] = len(h.points)
    sc["Simulation4UnkEnv/env_pts"] = pts; sc["Simulation4UnkEnv/env_n"] = npt
    sc["Simulation4UnkEnv/lidar_range"] = np.float64(1.5)
    np.savez_compressed(os.path.join(HERE, "pdf_scenarios.npz"), **sc)


if __name__ == "__main__":
    main()
