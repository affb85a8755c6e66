"""Dev tool: per-iteration / fixed cost and bench-batch launch time of the headline instantiation (N=8, 10 obstacles),
for A/B runs over variants/*.so (tools/build_variant.sh):  python tools/ab_iter.py [name ...]"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "humanoid-navigation-using-mpc-ldcbf_amd", "liblipmpc.so")
CHILD = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
import lipmpc
from importlib import import_module
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev = torch.device("cuda", 0); B = 4096; N = 8; n_obs = 10
def timeit(sv, args, out, reps=30):
    for _ in range(3): sv.plan_step_batch(*args, out=out)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): sv.plan_step_batch(*args, out=out)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
xy, nv = synth.synthetic_fields(8, n_obs, 0.5, 9.5, (0, 0), (10, 10), seed=1)
oxy = torch.as_tensor(np.repeat(xy[:1], B, 0), device=dev).contiguous(); onv = torch.as_tensor(np.repeat(nv[:1], B, 0), device=dev).contiguous()
goal = torch.tensor([[10., 10.]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
state = torch.zeros((B, 5), dtype=torch.float64, device=dev); foot = torch.ones((B,), dtype=torch.int8, device=dev)
ts = {}
for mi in (3, 8):
    sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=1, max_iter=mi), 0)
    ts[mi] = timeit(sv, (state, goal, foot, oxy, onv, None), sv.alloc_outputs(B))
it = (ts[8] - ts[3]) / 5
# the bench batch
xy, nv = synth.synthetic_fields(B, n_obs, 0.5, 9.5, (0, 0), (10, 10), seed=1234)
oxy = torch.as_tensor(xy, device=dev); onv = torch.as_tensor(nv, device=dev)
walker = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=1), 0)
delta = torch.zeros((B,), dtype=torch.float64, device=dev)
st, ft = synth.walk_states(walker, oxy, onv, goal, 30, seed=99, delta=delta)
sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5), 0)
out = sv.alloc_outputs(B, with_diag=True)
t = timeit(sv, (st, goal, ft, oxy, onv, delta), out, 50)
itn = out["iters"].cpu().numpy(); rn = out["diag"][:, 0].cpu().numpy()
fin = timeit(lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=1, tol_interior=1e-11), 0), (st, goal, ft, oxy, onv, delta), sv.alloc_outputs(B), 50)
print("%%-14s per-iteration %%.3f us  fixed %%.1f us | bench batch: launch %%.1f us (IPM only %%.1f), iters mean %%.2f max %%d, rounds max %%d" %% (%r, it, ts[3] - 3 * it, t, fin, itn.mean(), itn.max(), rn.max()))
'''
names = sys.argv[1:] or ["(current)"]
keep = LIB + ".keep"
shutil.copy(LIB, keep)
try:
    for n in names:
        shutil.copy(keep if n == "(current)" else os.path.join(ROOT, "variants", n + ".so"), LIB)
        subprocess.run([sys.executable, "-c", CHILD % (ROOT, n)], check=True)
finally:
    shutil.copy(keep, LIB); os.remove(keep)
