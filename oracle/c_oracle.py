"""ctypes loader of the C oracle (oracle/liblipmpc_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_lib = None


def _params_struct():
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    return importlib.import_module("humanoid-navigation-using-mpc-ldcbf_amd._lib").LipmpcParamsC


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liblipmpc_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(path)
        _lib.lipmpc_oracle_plan_step_batch.restype = C.c_int
        _lib.lipmpc_oracle_plan_step_batch.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 19 + [C.c_int]
        _lib.lipmpc_oracle_max_threads.restype = C.c_int
    return _lib


def plan_step_batch(params, state, goal, first_foot, obs_xy=None, obs_nv=None, delta=None, n_threads=1, bounds=None,
                    c_eta_in=None):
    """params: LipMpcParams (host mirror of struct lipmpc_params).  numpy in, dict of numpy out."""
    lib = load()
    cp = params.to_c()
    B, N, n_obs = state.shape[0], params.N, params.n_obs_max
    words = params.active_words
    f = lambda a, dt: None if a is None else np.ascontiguousarray(a, dtype=dt)
    state, goal, first_foot = f(state, np.float64), f(goal, np.float64), f(first_foot, np.int8)
    obs_xy, obs_nv, delta, bounds = f(obs_xy, np.float64), f(obs_nv, np.int32), f(delta, np.float64), f(bounds, np.float64)
    c_eta_in = f(c_eta_in, np.float64)
    out = dict(U=np.empty((B, N, 2)), X=np.empty((B, N + 1, 4)), theta=np.empty((B, N + 1)), omega=np.empty((B, N)),
               obj=np.empty(B), status=np.empty(B, np.int32), iters=np.empty(B, np.int32),
               active=np.zeros((B, words), np.uint64), working=np.zeros((B, words), np.uint64),
               c_eta=np.zeros((B, max(n_obs, 1), 4)), diag=np.zeros((B, 8)))
    p = lambda a: C.c_void_p(0 if a is None else a.ctypes.data)
    rc = lib.lipmpc_oracle_plan_step_batch(
        C.cast(C.byref(cp), C.c_void_p), B, p(state), p(goal), p(first_foot), p(delta), p(obs_xy), p(obs_nv),
        p(out["U"]), p(out["X"]), p(out["theta"]), p(out["omega"]), p(out["obj"]), p(out["status"]), p(out["iters"]),
        p(out["active"]), p(out["working"]), p(out["c_eta"]) if n_obs else p(None), p(out["diag"]), p(bounds), p(c_eta_in),
        int(n_threads))
    if rc != 0:
        raise RuntimeError(f"lipmpc_oracle_plan_step_batch failed ({rc})")
    if not n_obs:
        out["c_eta"] = out["c_eta"][:, :0]
    return out
