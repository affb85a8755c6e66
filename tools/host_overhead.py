import sys, time, torch
sys.path.insert(0, '/root/repo')
import lipmpc
dev = torch.device('cuda', 0)
for B in (4, 4096):
    P = lipmpc.LipMpcParams(N=8, n_obs_max=10, v_max=5)
    sv = lipmpc.BatchedLipMpc(P, 0)
    st = torch.zeros((B, 5), dtype=torch.float64, device=dev); goal = torch.ones((B, 2), dtype=torch.float64, device=dev) * 5
    foot = torch.ones((B,), dtype=torch.int8, device=dev)
    xy = torch.zeros((B, 10, 5, 2), dtype=torch.float64, device=dev); nv = torch.zeros((B, 10), dtype=torch.int32, device=dev)
    out = sv.alloc_outputs(B)
    for _ in range(10): sv.plan_step_batch(st, goal, foot, xy, nv, None, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): sv.plan_step_batch(st, goal, foot, xy, nv, None, out=out)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: host enqueue {1e6*(t1-t0)/500:.1f} us per call, with sync {1e6*(t2-t0)/500:.1f} us per call")
