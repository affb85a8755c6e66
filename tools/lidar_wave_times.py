"""Dev tool (LIPMPC_LIDAR_STOP=10: which share of the scans the chain proof clusters, by reading count).  LIPMPC_LIDAR_STOP=9: when every scan of the config-5 bench batch starts and ends (-DLIPMPC_LIDAR_PHASES variant, LIPMPC_LIDAR_STOP=9:
each wave records the 100 MHz wall clock at entry and exit), against its reading count: is the launch bound by its heaviest
waves' own length or by what shares their SIMD?"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lipmpc
from importlib import import_module
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
mode = os.environ.get("LIPMPC_LIDAR_STOP")
assert mode in ("9", "10")
dev = torch.device("cuda", 0); B = 4096
exy, env = synth.synthetic_fields(1, 20, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=9, delta=0.6)
rings = [exy[0, j, : env[0, j]] for j in range(20) if env[0, j] > 0]
sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, resolution=360, n_obs_max=12, v_max=32, device=0)
gen = torch.Generator(device=dev).manual_seed(3)
pos = torch.rand((B, 2), dtype=torch.float64, device=dev, generator=gen) * 7.0 - 1.0
state = torch.zeros((B, 5), dtype=torch.float64, device=dev); state[:, 0] = pos[:, 0]; state[:, 2] = pos[:, 1]
noise = 0.01 * torch.randn((B, 360, 2), dtype=torch.float64, device=dev, generator=gen)
o = sensor.sense(state, noise, with_debug=True, schedule=None); torch.cuda.synchronize()
npts = (~torch.isnan(o["hits"][:, :, 0])).sum(1).cpu().numpy()
sen = sensor.alloc_outputs(B, rings=False, c_eta=True)
sched = sensor.make_schedule(B)
for rep in range(3):
    sensor.sense(state, noise, out=sen, schedule=sched); torch.cuda.synchronize()
if mode == "10":
    route = sen["n_inferred"].cpu().numpy()
    print("clustered by chains: %.1f %% of the %d scans" % (100.0 * route.mean(), B))
    for lo, hi in ((0, 80), (80, 112), (112, 144), (144, 192), (192, 256), (256, 361)):
        m = (npts >= lo) & (npts < hi)
        if m.any():
            print("  readings %3d-%3d: %4d scans, %.1f %% by chains" % (lo, hi - 1, m.sum(), 100.0 * route[m].mean()))
    sys.exit(0)
t0 = sen["n_inferred"].cpu().numpy().astype(np.int64); t1 = sen["overflow"].cpu().numpy().astype(np.int64)
base = t0.min()
start, end = (t0 - base) / 100.0, (t1 - base) / 100.0            # us
dur = end - start
print("launch: last wave ends at %.1f us; waves start within %.1f us" % (end.max(), start.max()))
for lo, hi in ((0, 80), (80, 112), (112, 144), (144, 192), (192, 256), (256, 361)):
    m = (npts >= lo) & (npts < hi)
    if m.any():
        print("readings %3d-%3d: %4d waves, duration mean %.1f max %.1f us, end mean %.1f max %.1f us" % (lo, hi - 1, m.sum(), dur[m].mean(), dur[m].max(), end[m].mean(), end[m].max()))
late = np.argsort(-end)[:12]
print("latest waves (readings, start, duration, end):", [(int(npts[i]), round(start[i], 1), round(dur[i], 1), round(end[i], 1)) for i in late])
hits = o["hits"].cpu().numpy()
for i in late[:3]:                                   # what the slowest scans look like: pieces of consecutive readings within eps
    v = ~np.isnan(hits[i, :, 0]); pts = hits[i][v]
    d = np.hypot(*(pts[1:] - pts[:-1]).T)
    cuts = np.nonzero(d > 0.3)[0]
    sizes = np.diff(np.concatenate([[0], cuts + 1, [len(pts)]]))
    print("  scan with %d readings ending at %.1f us: pieces %s, extent %.2f x %.2f m, robot at (%.2f, %.2f)" % (
        len(pts), end[i], sizes.tolist(), np.ptp(pts[:, 0]), np.ptp(pts[:, 1]), pos[i, 0].item(), pos[i, 1].item()))
order = sched[2:2 + B].cpu().numpy()
posn = np.empty(B, np.int64); posn[order] = np.arange(B)
simd = posn % 1024
loads = np.zeros(1024); np.add.at(loads, simd, dur)
ends = np.zeros(1024); np.maximum.at(ends, simd, end)
print("per SIMD: sum of its four waves' durations mean %.1f max %.1f; last end mean %.1f max %.1f" % (loads.mean(), loads.max(), ends.mean(), ends.max()))
