"""Dev tool: config 4 (N = 16, 50 obstacles, B = 4096, bench.py's recipe) as the single dispatching kernel and as the split
launch (lipmpc_set_workspace): ms per step, class populations.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lipmpc  # noqa: E402
import importlib.util  # noqa: E402
from importlib import import_module  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev = torch.device("cuda", 0)
B, N, n_obs = int(os.environ.get("B", 4096)), 16, 50
inp = bench.make_inputs(lipmpc, synth, B, N, n_obs, 70000, 5, dev, 0, n_fields=512, walk_steps=20)
args = (inp["state"], inp["goal"], inp["foot"], inp["obs_xy"], inp["obs_nv"], inp["delta"])
P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5)
for name, auto in (("single kernel", False), ("split launch", True)):
    sv = lipmpc.BatchedLipMpc(P)
    sv.auto_workspace = auto
    out = sv.alloc_outputs(B)
    ms = bench._time_ms(lambda: sv.plan_step_batch(*args, out=out), reps=10)
    line = f"{name}: {ms:.4f} ms per {B} problems = {B / ms / 1e3:.3f} M solves/s"
    if auto:
        line += f"; class counts (1,2,4,13,25 slots) {sv._ws[:5].cpu().numpy().tolist()}"
    print(line, flush=True)
