"""Dev tool: what the fixed part of a launch (everything but the interior-point iterations) is made of: the bench batch
at max_iter = 1 with the ring front end, with the half-spaces given (no geometry), and on zero obstacles."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lipmpc
from importlib import import_module
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev = torch.device("cuda", 0); B = 4096; N = 8; n_obs = 10
xy, nv = synth.synthetic_fields(B, n_obs, 0.5, 9.5, (0, 0), (10, 10), seed=1234)
oxy = torch.as_tensor(xy, device=dev); onv = torch.as_tensor(nv, device=dev)
goal = torch.tensor([[10., 10.]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
walker = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=1), 0)
state, foot = synth.walk_states(walker, oxy, onv, goal, 30, seed=99)
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for mi in (1, 2):
    sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=1, max_iter=mi), 0)
    out = sv.alloc_outputs(B, with_c_eta=True)
    a = t(lambda: sv.plan_step_batch(state, goal, foot, oxy, onv, None, out=out))
    ce = out["c_eta"].clone(); out2 = sv.alloc_outputs(B)
    b = t(lambda: sv.plan_step_batch_c_eta(state, goal, foot, ce, None, out=out2))
    out3 = sv.alloc_outputs(B)
    c = t(lambda: sv.plan_step_batch(state, goal, foot, oxy, onv, None, out=out3))
    print(f"max_iter={mi}: rings + c_eta output {a:.1f} us, c_eta given {b:.1f} us, rings (no c_eta output) {c:.1f} us")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
x = torch.zeros(64, device=dev)
def empty(): x.add_(1)
print(f"back-to-back tiny torch kernel: {t(empty, 200):.2f} us")
