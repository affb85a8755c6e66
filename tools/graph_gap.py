"""Dev tool: what the gap between consecutive step launches costs on the bench batch -- K launches enqueued one by one
against the same K launches captured once in a HIP graph and replayed (torch.cuda.CUDAGraph)."""
import importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lipmpc
from importlib import import_module
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dev = torch.device("cuda", 0)
i = bench.make_inputs(lipmpc, synth, 4096, 8, 10, 0, 0, dev, 0)
sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=8, n_obs_max=10, v_max=5), 0)
out = sv.alloc_outputs(4096)
K = 20
step = lambda: sv.plan_step_batch(i["state"], i["goal"], i["foot"], i["obs_xy"], i["obs_nv"], i["delta"], out=out)
for _ in range(5): step()
torch.cuda.synchronize()
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
t_loop = timed(lambda: [step() for _ in range(K)])
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(K): step()
torch.cuda.synchronize()
t_graph = timed(lambda: g.replay())
print(f"{K} steps enqueued one by one: {t_loop / K * 1e3:.4f} ms per step; captured in one graph: {t_graph / K * 1e3:.4f} ms per step")
