"""Copies the summaries tools/profile_r04.sh left under gpurun_out/prof_<tag>_* into profiles/ (tracked) under per-round names and
merges their traffic records into profiles/traffic.json (what bench.py reads: HBM bytes and executed FP64 flops per launch of
the headline workload, of config 4 and of the closed-loop rollout).  Usage: python tools/collect_profiles.py r04"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
CLOCK_GHZ = 2.4


def cp(src, dst):
    if os.path.exists(src):
        shutil.copy(src, dst)
        print("copied", os.path.relpath(dst, ROOT))


traffic_path = os.path.join(P, "traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
d = os.path.join(G, f"prof_{tag}_final")
cp(os.path.join(d, "kernel_stats.csv"), os.path.join(P, f"{tag}_final_kernel_stats.csv"))
cp(os.path.join(d, "bench_launches.txt"), os.path.join(P, f"{tag}_final_bench_launches.txt"))
cp(os.path.join(d, "pmc.csv"), os.path.join(P, f"{tag}_final_pmc.csv"))
if os.path.exists(os.path.join(d, "bench_stats.log")):
    line = [ln for ln in open(os.path.join(d, "bench_stats.log")) if ln.startswith("{")]
    if line:
        open(os.path.join(P, f"{tag}_final_bench_under_rocprof.json"), "w").write(line[-1])
if os.path.exists(os.path.join(d, "traffic.json")):
    t = json.load(open(os.path.join(d, "traffic.json")))
    c = {r[0]: float(r[1]) for r in list(csv.reader(open(os.path.join(d, "pmc.csv"))))[1:]}
    us = float(open(os.path.join(d, "bench_launches.txt")).read().split("mean ")[1].split(" us")[0])
    for k, v in t.items():
        if not k.startswith("_"):
            v["wave_alive_fraction"] = c.get("SQ_WAVE_CYCLES", 0) * 4 / (us * 1e-6 * CLOCK_GHZ * 1e9 * 1024)
            v["trace_us_per_launch"] = us
            traffic[k] = v
    traffic["_note"] = t.get("_note", traffic.get("_note"))
for wl, rename in (("cfg4", None), ("rollout", None), ("cfg5_scan", "cfg5_scan_B4096"), ("cfg5_solve", "cfg5_solve_B4096")):
    d = os.path.join(G, f"prof_{tag}_{wl}")
    for f in ("kernel_stats.csv", "launches.txt", "pmc.csv"):
        cp(os.path.join(d, f), os.path.join(P, f"{tag}_{wl}_{f}"))
    tp = os.path.join(d, "traffic.json")
    if os.path.exists(tp):
        for k, v in json.load(open(tp)).items():
            v = dict(v)
            # the span of one workload launch: the slowest matching kernel of the trace (the kernels of a split launch overlap)
            spans = [float(ln.split("mean ")[1].split(" us")[0]) for ln in open(os.path.join(d, "launches.txt")) if "mean " in ln]
            if wl == "cfg4":
                v["trace_us_slowest_kernel"] = max(spans)
            else:
                v["trace_us_per_launch"] = max(spans)
            slots = min(v.get("sq_waves", 0) or 1024, 1024 * (4 if wl == "cfg5_scan" else 1))     # the scan runs four waves per SIMD
            span_us = (v.get("workload", {}).get("ms", 0) * 1e3) if wl == "cfg4" else max(spans)
            if v.get("sq_wave_cycles") and span_us:
                v["wave_alive_fraction"] = v["sq_wave_cycles"] * 4 / (span_us * 1e-6 * CLOCK_GHZ * 1e9 * slots)
            traffic[rename or k] = v
json.dump(traffic, open(traffic_path, "w"), indent=1)
print("wrote profiles/traffic.json:", [k for k in traffic if not k.startswith("_")])
