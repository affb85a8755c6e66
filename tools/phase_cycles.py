"""Dev tool: where a step launch spends its time, section by section and WAVE by wave (the -DLIPMPC_PHASE_TIMING variant:
tools/build_variant.sh phase "-DLIPMPC_PHASE_TIMING" "api 16_5 32_25 L:32_1 L:32_2 L:32_4";
LIPMPC_LIB=variants/phase.so LIPMPC_ALLOW_VARIANT=1 python tools/phase_cycles.py [u8 u16 bench cfg4]).
Every wave books constant-clock time per section in LDS, once per wave pass, whatever subset of its groups is still running
(lipmpc_kernel.hpp: ph_mark), plus the part of each section spent with ONE group alive -- the tail inside the wave.  Printed:
mean per wave, per iteration / round, the share of the launch's wave-time that is single-group time, and the slowest waves."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LIPMPC_ALLOW_VARIANT", "1")
import lipmpc  # noqa: E402
from importlib import import_module  # noqa: E402
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
solver_mod = import_module("humanoid-navigation-using-mpc-ldcbf_amd.solver")
_ptr = solver_mod._ptr
NAMES = ["iteration head", "reciprocals + K", "factorisation", "predictor rhs + solve", "predictor rows/ratio", "corrector rhs + solve",
         "corrector rows/update", "finish K + factor", "finish equality solve", "finish ratio/exchange", "front end", "outputs"]
dev = torch.device("cuda", 0)
PH_WORDS = 32


def run(tag, N, n_obs, state, goal, foot, obs_xy, obs_nv, delta, flags=0, max_iter=60, split=True):
    B = state.shape[0]
    sv = lipmpc.BatchedLipMpc(lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, flags=flags, max_iter=max_iter), 0)
    assert sv.lib.lipmpc_version() >= 1000, "load the -DLIPMPC_PHASE_TIMING variant (LIPMPC_LIB=variants/phase.so)"
    sv.auto_workspace = split
    sv._ensure_workspace(B)
    out = sv.alloc_outputs(B)
    ph = torch.zeros((B, PH_WORDS), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2):
        ph.zero_()
        rc = sv.lib.lipmpc_plan_step_batch(sv._h, B, _ptr(state), _ptr(goal), _ptr(foot), _ptr(delta), _ptr(obs_xy), _ptr(obs_nv),
                                           _ptr(out["U"]), _ptr(out["X"]), _ptr(out["theta"]), _ptr(out["omega"]), _ptr(out["obj"]),
                                           _ptr(out["status"]), _ptr(out["iters"]), _ptr(out["active"]), None, None, _ptr(ph), None,
                                           C.c_void_p(stream))
        assert rc == 0
    torch.cuda.synchronize()
    p = ph.cpu().numpy()
    w = p[p[:, 28] == 1.0]                              # one record per wave, at the wave's first problem
    it_w, rd_w, life = w[:, 26], w[:, 27], w[:, 24]
    sec, solo = w[:, :12], w[:, 12:24]
    tot = sec.sum(1)
    ghz = np.median(w[:, 25] / np.maximum(life, 1.0))
    print(f"== {tag}: {len(w)} waves, mean wave {life.mean() / 1e3:.1f} us, slowest {life.max() / 1e3:.1f} us (sections cover {tot.sum() / life.sum():.3f} of it); "
          f"shader clock counter {ghz:.3f} ticks/ns; wave-max iterations mean {it_w.mean():.1f}, rounds mean {rd_w.mean():.2f}; "
          f"single-group share of all wave time {solo.sum() / tot.sum():.3f}")
    for k in range(12):
        per = ""
        if k <= 6:
            per = f"  = {sec[:, k].sum() / max(it_w.sum(), 1):8.0f} ns per wave iteration"
        elif k <= 9:
            per = f"  = {sec[:, k].sum() / max(rd_w.sum(), 1):8.0f} ns per wave round"
        print(f"   {k:2d} {NAMES[k]:26s} {sec[:, k].mean() / 1e3:9.2f} us/wave ({100 * sec[:, k].sum() / tot.sum():5.1f} %), single-group part {100 * solo[:, k].sum() / max(sec[:, k].sum(), 1):5.1f} %{per}")
    print(f"   per wave iteration {sec[:, :7].sum() / max(it_w.sum(), 1) / 1e3:.2f} us, per wave round {sec[:, 7:10].sum() / max(rd_w.sum(), 1) / 1e3:.2f} us, "
          f"fixed (front end + outputs) {sec[:, 10:].sum(1).mean() / 1e3:.2f} us")
    print(f"   inside 'front end': kernel entry -> headings done {w[:, 30].mean() / 1e3:.2f} us, -> front end done {w[:, 29].mean() / 1e3:.2f} us, "
          f"-> first iteration {sec[:, 10].mean() / 1e3:.2f} us")
    for i in np.argsort(-life)[:6]:                     # the waves the launch waits for
        print(f"   slow wave: {life[i] / 1e3:7.1f} us  iterations {it_w[i]:3.0f} x {sec[i, :7].sum() / max(it_w[i], 1) / 1e3:5.2f} us  rounds {rd_w[i]:2.0f} x "
              f"{sec[i, 7:10].sum() / max(rd_w[i], 1) / 1e3:5.2f} us (K+factor {sec[i, 7] / 1e3:.1f}, equality solve {sec[i, 8] / 1e3:.1f}, ratio/exchange {sec[i, 9] / 1e3:.1f})  "
              f"front {sec[i, 10] / 1e3:.1f} out {sec[i, 11] / 1e3:.1f}; single-group time {solo[i].sum() / 1e3:.1f} us")
    return w


def uniform(N, n_obs, B=4096):
    hi, g = (9.5, 10.0) if N <= 8 else (15.5, 16.0)
    xy, nv = synth.synthetic_fields(4, max(n_obs, 1), 0.5, hi, (0, 0), (g, g), seed=1)
    obs_xy = torch.as_tensor(np.repeat(xy[:1, :n_obs], B, 0), device=dev).contiguous() if n_obs else None
    obs_nv = torch.as_tensor(np.repeat(nv[:1, :n_obs], B, 0), device=dev).contiguous() if n_obs else None
    goal = torch.tensor([[g, g]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    state = torch.zeros((B, 5), dtype=torch.float64, device=dev)
    foot = torch.ones((B,), dtype=torch.int8, device=dev)
    run(f"uniform N={N} n_obs={n_obs}", N, n_obs, state, goal, foot, obs_xy, obs_nv, None)


if __name__ == "__main__":
    which = sys.argv[1:] or ["u8", "u16", "bench", "cfg4"]
    if "u8" in which:
        uniform(8, 10)
    if "u16" in which:
        uniform(16, 0)
        uniform(16, 50)
    if "bench" in which or "cfg4" in which:
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        if "bench" in which:
            i = bench.make_inputs(lipmpc, synth, 4096, 8, 10, 0, 0, dev, 0)
            run("bench batch (N=8, 10 obstacles)", 8, 10, i["state"], i["goal"], i["foot"], i["obs_xy"], i["obs_nv"], i["delta"])
        if "cfg4" in which:
            i = bench.make_inputs(lipmpc, synth, 4096, 16, 50, 70000, 5, dev, 0, n_fields=512, walk_steps=20)
            run("config 4 batch (N=16, 50 obstacles)", 16, 50, i["state"], i["goal"], i["foot"], i["obs_xy"], i["obs_nv"], i["delta"])
