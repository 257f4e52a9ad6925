"""ctypes binding of libpgf_hip.so (C ABI in include/pgf_hip.h).

There is NO CPU fallback: importing this module without the built library, or
calling any compute entry without a GPU, raises.  Status codes are mapped to the
reference's exception convention: singular / inertia ->
``LinearSolverError``, invalid argument -> ``ValueError``, anything else ->
``RuntimeError``.
"""

from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from .errors import LinearSolverError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpgf_hip.so")

PGF_OK, PGF_SINGULAR, PGF_INERTIA, PGF_INVALID, PGF_NOT_READY, PGF_HIP_ERROR = 0, 1, 2, 3, 4, 100
PGF_HOST, PGF_DEVICE = 0, 1
STEP_RECOMPUTE_MASK, STEP_REFACTOR, STEP_REFACTOR_ON_CHANGE = 1, 2, 4
CREATE_SPARSE = 1

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_ip = C.POINTER(C.c_int)
_h = C.c_void_p

# name -> (restype, argtypes); must list EVERY symbol include/pgf_hip.h declares
SIGNATURES = {
    "pgf_version": (C.c_int, []),
    "pgf_device_count": (C.c_int, [_ip]),
    "pgf_last_error": (C.c_char_p, [_h]),
    "pgf_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(_h)]),
    "pgf_destroy": (C.c_int, [_h]),
    "pgf_set_bounds": (C.c_int, [_h, _dp, _dp]),
    "pgf_set_outer": (C.c_int, [_h, _dp, _dp, C.c_double, C.c_double]),
    "pgf_set_derivs_dense": (C.c_int, [_h, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]),
    "pgf_set_derivs_csr": (C.c_int, [_h, _ip, _ip, _dp, _ip, _ip, _dp]),
    "pgf_active_set": (C.c_int, [_h, _dp, _dp, C.c_double, _u8p]),
    "pgf_set_active_set": (C.c_int, [_h, _u8p]),
    "pgf_factor": (C.c_int, [_h, _ip]),
    "pgf_newton_solve": (C.c_int, [_h, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "pgf_residual": (C.c_int, [_h, _dp, _dp, _dp, _dp, _u8p, _dp]),
    "pgf_linear_solve": (C.c_int, [_h, _dp, C.c_int, _dp]),
    "pgf_reduced_dims": (C.c_int, [_h, _ip, _ip]),
    "pgf_get_kkt": (C.c_int, [_h, _dp, C.c_int64]),
    "pgf_sparse_set_pattern": (C.c_int, [_h, C.c_int, _ip, C.c_int, _ip, _ip, _ip, _ip, C.c_int, _ip, _ip, _ip,
                                         _ip, _ip, _ip]),
    "pgf_sparse_set_values": (C.c_int, [_h, _dp, _dp]),
    "pgf_qp_set_vectors": (C.c_int, [_h, _dp, _dp]),
    "pgf_qp_set_problem": (C.c_int, [_h, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_int]),
    "pgf_qp_set_point": (C.c_int, [_h, _dp, _dp]),
    "pgf_qp_get_point": (C.c_int, [_h, _dp, _dp]),
    "pgf_qp_get_mask": (C.c_int, [_h, _u8p]),
    "pgf_qp_update_active_set": (C.c_int, [_h, C.c_double, _ip]),
    "pgf_qp_advance_outer": (C.c_int, [_h, C.c_double, C.c_double]),
    "pgf_qp_step": (C.c_int, [_h, C.c_uint, C.c_double, C.c_int, _ip, _dp]),
    "pgf_qp_step_async": (C.c_int, [_h, C.c_uint, C.c_double]),
    "pgf_qp_sync": (C.c_int, [_h, _ip, _dp]),
    "pgf_qp_residual_norm": (C.c_int, [_h, _dp, C.c_void_p]),
    "pgf_qp_measures": (C.c_int, [_h, C.c_double, _dp]),
    "pgf_stream": (C.c_int, [_h, C.POINTER(C.c_void_p)]),
    "pgf_profile_enable": (C.c_int, [_h, C.c_int]),
    "pgf_profile_read": (C.c_int, [_h, _dp, C.POINTER(C.c_int64), _dp, _dp]),
    "pgf_batch_create": (C.c_int, [C.POINTER(_h), C.c_int, C.POINTER(_h)]),
    "pgf_batch_destroy": (C.c_int, [_h]),
    "pgf_batch_last_error": (C.c_char_p, [_h]),
    "pgf_batch_advance_outer": (C.c_int, [_h, C.c_double, C.c_double]),
    "pgf_batch_advance_outer_each": (C.c_int, [_h, _dp, _dp, _u8p]),
    "pgf_batch_set_frozen": (C.c_int, [_h, _u8p]),
    "pgf_batch_update_active_set": (C.c_int, [_h, C.c_double]),
    "pgf_batch_step_async": (C.c_int, [_h, C.c_uint, C.c_double]),
    "pgf_batch_sync": (C.c_int, [_h, _ip, _ip, _dp]),
    "pgf_batch_residual_norms": (C.c_int, [_h, _dp, C.c_void_p]),
    "pgf_batch_get_points": (C.c_int, [_h, _dp, _dp]),
    "pgf_batch_get_masks": (C.c_int, [_h, _u8p]),
    "pgf_batch_measures": (C.c_int, [_h, C.c_double, _dp]),
    "pgf_batch_stream": (C.c_int, [_h, C.POINTER(C.c_void_p)]),
    "pgf_batch_profile_enable": (C.c_int, [_h, C.c_int]),
    "pgf_batch_profile_read": (C.c_int, [_h, _dp, C.POINTER(C.c_int64), _dp]),
    "pgf_ls_create_dense": (C.c_int, [C.c_int, _dp, C.c_int64, C.c_int, C.c_int, C.POINTER(_h)]),
    "pgf_ls_solve": (C.c_int, [_h, _dp, C.c_int, _dp]),
    "pgf_ls_num_neg": (C.c_int, [_h, _ip]),
    "pgf_ls_get_factor": (C.c_int, [_h, _dp, C.c_int64]),
    "pgf_ls_destroy": (C.c_int, [_h]),
    "pgf_bench_update": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "pgf_profile_read_ex": (C.c_int, [_h, _dp, C.c_int]),
    "pgf_batch_ctl_init": (C.c_int, [_h, C.c_double, C.c_double, _dp, C.c_int]),
    "pgf_batch_ctl_iterate": (C.c_int, [_h, C.c_uint, C.c_double, C.c_int]),
    "pgf_batch_ctl_read": (C.c_int, [_h, _dp, C.POINTER(C.c_uint8), _dp, C.c_int]),
    "pgf_kkt_apply": (C.c_int, [_h, _dp, _dp]),
    "pgf_set_refinement": (C.c_int, [_h, C.c_int, C.c_double, C.c_double]),
    "pgf_refinement_stats": (C.c_int, [_h, _ip, _ip, _dp]),
    "pgf_debug_fail_next_chain": (C.c_int, [_h]),
    "pgf_debug_chain_enable": (C.c_int, [C.c_int]),
    "pgf_debug_fail_next_helper": (C.c_int, [_h]),
    "pgf_debug_chain_helpers": (C.c_int, [C.c_int]),
    "pgf_debug_factor_kind": (C.c_int, [_h]),
    "pgf_batch_refinement_stats": (C.c_int, [_h, _ip]),
    "pgf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pgf_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "pgf_comm_destroy": (C.c_int, [C.c_void_p]),
    "pgf_batch_allgather_norms": (C.c_int, [_h, C.c_void_p, C.c_void_p]),
    "pgf_batch_debug_fail_next_helper": (C.c_int, [_h]),
}

_lib = None


def load():
    """Load the shared library (no GPU needed for loading)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m pygradflow_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.  If this
    # library pulled in the system copy first, a later `import torch` would initialise a
    # second runtime that finds "No HIP GPUs".  Importing torch first (when it is installed)
    # makes both share torch's copy; without torch the system runtime is used.
    if "torch" not in sys.modules and not os.environ.get("PGF_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def require_gpu():
    lib = load()
    cnt = C.c_int(0)
    rc = lib.pgf_device_count(C.byref(cnt))
    if rc != PGF_OK or cnt.value < 1:
        raise RuntimeError(
            "pygradflow_amd needs an AMD GPU (gfx950) visible to HIP; none found "
            f"(status {rc}). There is no CPU fallback."
        )
    return cnt.value


def dptr(a):
    """double* of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def u8ptr(a):
    if a is None:
        return None
    assert a.dtype in (np.uint8, np.bool_) and a.flags.c_contiguous
    return a.ctypes.data_as(_u8p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def check(rc, handle=None, what="", batch=None):
    if rc == PGF_OK:
        return
    msg = ""
    if batch is not None:
        msg = load().pgf_batch_last_error(batch).decode(errors="replace")
    elif handle is not None:
        msg = load().pgf_last_error(handle).decode(errors="replace")
    text = f"{what}: {msg}" if what else msg
    if rc in (PGF_SINGULAR, PGF_INERTIA):
        raise LinearSolverError(text or ("singular matrix" if rc == PGF_SINGULAR else "Invalid matrix inertia"))
    if rc == PGF_INVALID:
        raise ValueError(text or "invalid argument")
    raise RuntimeError(f"{text} (status {rc})")
