"""Dev tool: where the hardware dispatcher places the scans.  With the -DLIPMPC_LIDAR_PHASES variant (LIPMPC_LIB) and
LIPMPC_LIDAR_STOP=8 every wave of lidar_sense_kernel records HW_ID / XCC_ID for its launch position and stays resident for a
while: prints how launch positions map to (XCD, SE, CU, SIMD) -- which positions share a SIMD."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lipmpc
from importlib import import_module
synth = import_module("humanoid-navigation-using-mpc-ldcbf_amd.synth")
assert os.environ.get("LIPMPC_LIDAR_STOP") == "8"
dev = torch.device("cuda", 0); B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
exy, env = synth.synthetic_fields(1, 20, -1.0, 6.0, (-5.0, -5.0), (50.0, 50.0), seed=9, delta=0.6)
rings = [exy[0, j, : env[0, j]] for j in range(20) if env[0, j] > 0]
sensor = lipmpc.LidarSensor(rings, lidar_range=1.5, resolution=360, n_obs_max=12, v_max=32, device=0)
state = torch.zeros((B, 5), dtype=torch.float64, device=dev)
noise = torch.zeros((B, 360, 2), dtype=torch.float64, device=dev)
sen = sensor.alloc_outputs(B, rings=False, c_eta=True)
for rep in range(3):
    sensor.sense(state, noise, out=sen, schedule=None); torch.cuda.synchronize()
    hw = sen["n_inferred"].cpu().numpy().astype(np.uint32); xcc = sen["overflow"].cpu().numpy().astype(np.uint32) & 0xF
    wave, simd, pipe, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = key * 4 + simd
    print(f"run {rep}: {len(np.unique(xcc))} XCDs, {len(np.unique(key))} CUs, {len(np.unique(simd_key))} SIMDs in use; waves per SIMD min/max "
          f"{np.bincount(np.unique(simd_key, return_inverse=True)[1]).min()}/{np.bincount(np.unique(simd_key, return_inverse=True)[1]).max()}")
    print("  first 40 positions (xcd, se, cu, simd, wave):", [(int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i]), int(wave[i])) for i in range(24)])
    # positions sharing the SIMD of position 0 / 1 / 8
    for p0 in (0, 1, 8, 9):
        print(f"  positions on the SIMD of position {p0}:", np.nonzero(simd_key == simd_key[p0])[0].tolist()[:24])
    for p0 in (0, 8):
        print(f"  positions on the CU of position {p0}:", np.nonzero(key == key[p0])[0].tolist()[:40])
    # is xcd = position % 8 ?
    print("  xcd == position % 8 for all:", bool((xcc == (np.arange(B) % 8)).all()), " distinct xcd ids:", np.unique(xcc).tolist())
    # period structure: for each position p, the next position on the same SIMD
    nxt = []
    order = np.argsort(simd_key, kind="stable")
    sk = simd_key[order]
    for a, b_ in zip(order[:-1], order[1:]):
        if simd_key[a] == simd_key[b_]:
            nxt.append(b_ - a)
    vals, cnt = np.unique(nxt, return_counts=True)
    print("  gaps between consecutive positions on one SIMD (gap: count):", {int(v): int(c) for v, c in zip(vals, cnt)})
np.save(os.path.join(ROOT, "gpurun_out", "lidar_placement_simd.npy"), simd_key)
