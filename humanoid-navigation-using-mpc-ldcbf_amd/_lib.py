"""ctypes binding of liblipmpc.so (C ABI: include/lipmpc.h).  There is no CPU fallback: if the
HIP library is missing this module raises at import of the solver."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblipmpc.so")


class LipmpcParamsC(C.Structure):
    """struct lipmpc_params (include/lipmpc.h)"""
    _fields_ = [
        ("N", C.c_int32), ("n_obs_max", C.c_int32), ("v_max", C.c_int32), ("max_iter", C.c_int32),
        ("flags", C.c_int32), ("finish_rounds", C.c_int32),
        ("dt", C.c_double), ("g", C.c_double), ("h_com", C.c_double), ("alpha", C.c_double),
        ("l_max", C.c_double * 2), ("l_min", C.c_double * 2), ("v_min", C.c_double * 2),
        ("v_max_xy", C.c_double * 2),
        ("omega_max", C.c_double), ("ell", C.c_double), ("sampling_time", C.c_double),
        ("tol", C.c_double), ("tol_interior", C.c_double), ("k0_tol", C.c_double),
    ]


EXPORTS = ("lipmpc_default_params", "lipmpc_create", "lipmpc_destroy", "lipmpc_num_rows",
           "lipmpc_active_words", "lipmpc_plan_step_batch", "lipmpc_plan_step_batch_c_eta", "lipmpc_advance_batch", "lipmpc_fleet_update_batch", "lipmpc_rollout_batch", "lipmpc_lidar_sense_batch", "lipmpc_lidar_c_eta_batch", "lipmpc_lidar_schedule_words", "lipmpc_sense_plan_step_batch", "lipmpc_set_schedule", "lipmpc_schedule_words", "lipmpc_set_workspace", "lipmpc_workspace_bytes",
           "lipmpc_strerror", "lipmpc_version")

ABI_VERSION = 5          # LIPMPC_ABI_VERSION of include/lipmpc.h this binding is written for
VARIANT_BASE = 1000      # LIPMPC_VARIANT_BASE: instrumented development builds report ABI_VERSION + this
DIAG_WORDS = 8           # LIPMPC_DIAG_WORDS
TIGHT_TOL = 1e-7         # LIPMPC_TIGHT_TOL

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # LIPMPC_LIB: another build of the same library (dev tools only: the instrumented / historical variants of tools/*.sh
    # are loaded from their own path instead of being copied over the shipped file)
    path = os.environ.get("LIPMPC_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C <package>/csrc)")
    lib = C.CDLL(path)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    # The argument lists bound below are those of ONE ABI version: a stale or historical build (LIPMPC_LIB) would take every
    # pointer after an inserted argument shifted by one, and an instrumented variant (version >= LIPMPC_VARIANT_BASE) writes
    # other buffer shapes.  Refuse both here, before any pointer is handed over; tools that drive a variant on purpose set
    # LIPMPC_ALLOW_VARIANT=1 and hand in the buffers that variant expects.
    lib.lipmpc_version.argtypes = []
    lib.lipmpc_version.restype = i32
    ver = int(lib.lipmpc_version())
    if ver != ABI_VERSION and not (ver == ABI_VERSION + VARIANT_BASE and os.environ.get("LIPMPC_ALLOW_VARIANT") == "1"):
        raise RuntimeError(f"{path}: lipmpc_version() = {ver}, this binding is written for ABI {ABI_VERSION} "
                           f"(include/lipmpc.h); rebuild the library (make -C <package>/csrc)")
    lib.lipmpc_default_params.argtypes = [C.POINTER(LipmpcParamsC)]
    lib.lipmpc_default_params.restype = i32
    lib.lipmpc_create.argtypes = [C.POINTER(LipmpcParamsC), i32, C.POINTER(vp)]
    lib.lipmpc_create.restype = i32
    lib.lipmpc_destroy.argtypes = [vp]
    lib.lipmpc_destroy.restype = None
    lib.lipmpc_num_rows.argtypes = [C.POINTER(LipmpcParamsC)]
    lib.lipmpc_num_rows.restype = i64
    lib.lipmpc_active_words.argtypes = [C.POINTER(LipmpcParamsC)]
    lib.lipmpc_active_words.restype = i64
    lib.lipmpc_plan_step_batch.argtypes = [vp, i64] + [vp] * 19
    lib.lipmpc_plan_step_batch.restype = i32
    lib.lipmpc_plan_step_batch_c_eta.argtypes = [vp, i64] + [vp] * 18
    lib.lipmpc_plan_step_batch_c_eta.restype = i32
    lib.lipmpc_advance_batch.argtypes = [vp, i64] + [vp] * 6
    lib.lipmpc_advance_batch.restype = i32
    lib.lipmpc_fleet_update_batch.argtypes = [vp, i64, C.c_int32, C.c_double] + [vp] * 17
    lib.lipmpc_fleet_update_batch.restype = i32
    lib.lipmpc_rollout_batch.argtypes = [vp, i64, C.c_int32, C.c_int32, C.c_double] + [vp] * 13
    lib.lipmpc_rollout_batch.restype = i32
    lib.lipmpc_lidar_sense_batch.argtypes = ([i32, i64] + [C.c_int32] * 4 + [C.c_double, C.c_double] + [C.c_int32] * 3
                                             + [vp] * 12)
    lib.lipmpc_lidar_sense_batch.restype = i32
    lib.lipmpc_lidar_c_eta_batch.argtypes = ([i32, i64] + [C.c_int32] * 4 + [C.c_double, C.c_double] + [C.c_int32] * 3
                                             + [vp] * 14)
    lib.lipmpc_lidar_c_eta_batch.restype = i32
    lib.lipmpc_lidar_schedule_words.argtypes = [i64]
    lib.lipmpc_lidar_schedule_words.restype = i64
    lib.lipmpc_sense_plan_step_batch.argtypes = ([vp, i64] + [C.c_int32] * 4 + [C.c_double, C.c_double, C.c_int32] + [vp] * 24)
    lib.lipmpc_sense_plan_step_batch.restype = i32
    lib.lipmpc_set_schedule.argtypes = [vp, vp, i64]
    lib.lipmpc_set_schedule.restype = i32
    lib.lipmpc_schedule_words.argtypes = [i64]
    lib.lipmpc_schedule_words.restype = i64
    lib.lipmpc_set_workspace.argtypes = [vp, vp, i64]
    lib.lipmpc_set_workspace.restype = i32
    lib.lipmpc_workspace_bytes.argtypes = [vp, i64]
    lib.lipmpc_workspace_bytes.restype = i64
    lib.lipmpc_strerror.argtypes = [i32]
    lib.lipmpc_strerror.restype = C.c_char_p
    lib.lipmpc_version.argtypes = []
    lib.lipmpc_version.restype = i32
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().lipmpc_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} ({code})")
