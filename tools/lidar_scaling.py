"""Dev tool: LiDAR scan kernel time against the number of robots (1024 robots = one wave per SIMD: the latency of a lone
wave; 2048+ = two co-resident waves per SIMD) on the config-5 bench map."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lipmpc
from importlib import import_module
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev = torch.device("cuda", 0)
exy, env = synth.synthetic_fields(1, 20, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=9, delta=0.6)
rings = [exy[0, j, : env[0, j]] for j in range(20) if env[0, j] > 0]
sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, resolution=360, n_obs_max=12, v_max=32, device=0)
for B in (256, 1024, 2048, 4096, 8192, 16384):
    gen = torch.Generator(device=dev).manual_seed(3)
    pos = torch.rand((B, 2), dtype=torch.float64, device=dev, generator=gen) * 7.0 - 1.0
    state = torch.zeros((B, 5), dtype=torch.float64, device=dev); state[:, 0] = pos[:, 0]; state[:, 2] = pos[:, 1]
    noise = 0.01 * torch.randn((B, 360, 2), dtype=torch.float64, device=dev, generator=gen)
    sen = sensor.alloc_outputs(B)
    for _ in range(3): sensor.sense(state, noise, out=sen)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): sensor.sense(state, noise, out=sen)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"B={B:6d}: scan {ms*1e3:8.1f} us  ({B/ms*1e-3:.2f} M scans/s)  mean inferred {float(sen['n_inferred'].double().mean()):.2f}")
