"""Dump the inputs (and the GPU's answers) of the config-4 parity batch and of bench.py's headline batch to
gpurun_out/*.npz, for offline work on the finish / active sets with the oracles (the states of these batches come from an
on-device closed-loop warm-up, so they can only be produced on the GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lipmpc  # noqa: E402
from test_gpu_configs import _walked_batch  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)


def dump(tag, B, N, n_obs, hi, goal, seed, steps, n_fields=None, delta_mix=False, rounds=0):
    b = _walked_batch(B, N, n_obs, hi, goal, seed=seed, max_steps=steps, n_fields=n_fields, delta_mix=delta_mix)
    P = lipmpc.LipMpcParams(N=N, n_obs_max=n_obs, v_max=5, finish_rounds=rounds)
    out = lipmpc.BatchedLipMpc(P).plan_step_batch(b["state"], b["goal"], b["foot"], b["obs_xy"], b["obs_nv"], b["delta"],
                                                  with_diag=True, with_c_eta=True)
    torch.cuda.synchronize()
    g = {"g_" + k: v.cpu().numpy() for k, v in out.items()}
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", tag + ".npz"), state=b["state"].cpu().numpy(), foot=b["foot"].cpu().numpy(),
                        goal=b["goal"].cpu().numpy(), delta=b["delta"].cpu().numpy(), xy=b["xy"].astype(np.float64), nv=b["nv"], **g)
    print(tag, np.bincount(g["g_status"], minlength=5), "iters", g["g_iters"].mean())


if __name__ == "__main__":
    dump("cfg4_batch", 4096, 16, 50, 15.5, (16.0, 16.0), 31, 20, n_fields=1024)
    dump("cfg2_batch", 4096, 8, 10, 9.5, (10.0, 10.0), 1234, 30, delta_mix=True)
