"""Dev tool: the config-5 closed loop of bench.py (4096 robots, 30 samples, one HIP graph per sample) on its own, for
rocprofv3 --kernel-trace --stats (tools/profile_fleet.sh)."""
import sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lipmpc
from importlib import import_module
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev = torch.device("cuda", 0); B, N, K = 4096, 3, 30
exy, env = synth.synthetic_fields(1, 20, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=9, delta=0.6)
rings = [exy[0, j, : env[0, j]] for j in range(20) if env[0, j] > 0]
gen = torch.Generator(device=dev).manual_seed(3)
goal = torch.tensor([[5.0, 5.0]], dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
foot = torch.ones((B,), dtype=torch.int8, device=dev)
fleet = lipmpc.UnknownEnvFleet(rings, N_horizon=N, lidar_range=1.5, resolution=360, n_obs_max=12, v_max=32, device=0)
st0 = torch.zeros((B, 5), dtype=torch.float64, device=dev)
st0[:, 0] = -1.8 + 0.5 * torch.rand((B,), dtype=torch.float64, device=dev, generator=gen)
st0[:, 2] = -1.5 + 7.5 * torch.rand((B,), dtype=torch.float64, device=dev, generator=gen)
fleet.run(st0, goal, foot, 3); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); r = fleet.run(st0, goal, foot, K, noise_seed=5); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"fleet {K} samples: {dt*1e3:.2f} ms, {int(r['n_steps'].sum())/dt/1e6:.2f} M robot-steps/s")
