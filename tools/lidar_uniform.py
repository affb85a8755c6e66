"""Dev tool: scan time of uniform batches (every robot at the same spot) for a few spots of the config-5 bench map:
how the scan's latency (256 robots: lone waves) and throughput (4096) depend on the number of readings."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lipmpc
from importlib import import_module
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
dev = torch.device("cuda", 0)
exy, env = synth.synthetic_fields(1, 20, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=9, delta=0.6)
rings = [exy[0, j, : env[0, j]] for j in range(20) if env[0, j] > 0]
sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, resolution=360, n_obs_max=12, v_max=32, device=0)
gen = torch.Generator(device=dev).manual_seed(3)
cand = torch.rand((64, 2), dtype=torch.float64, device=dev, generator=gen) * 7.0 - 1.0
st = torch.zeros((64, 5), dtype=torch.float64, device=dev); st[:, 0] = cand[:, 0]; st[:, 2] = cand[:, 1]
nz = 0.01 * torch.randn((64, 360, 2), dtype=torch.float64, device=dev, generator=gen)
o = sensor.sense(st, nz, with_debug=True); torch.cuda.synchronize()
npts = (~torch.isnan(o["hits"][:, :, 0])).sum(1).cpu().numpy(); ninf = o["n_inferred"].cpu().numpy()
order = np.argsort(npts)
for idx in order[[0, 16, 32, 48, 56, 60, 63]]:
    for B in (256, 4096):
        state = st[idx:idx + 1].repeat(B, 1).contiguous(); noise = nz[idx:idx + 1].repeat(B, 1, 1).contiguous()
        sen = sensor.alloc_outputs(B)
        for _ in range(3): sensor.sense(state, noise, out=sen)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): sensor.sense(state, noise, out=sen)
        e1.record(); torch.cuda.synchronize()
        print(f"readings {npts[idx]:3d} hulls {ninf[idx]}  B={B:5d}: {e0.elapsed_time(e1)/10*1e3:7.1f} us  (stop {os.environ.get('LIPMPC_LIDAR_STOP','0')})")
