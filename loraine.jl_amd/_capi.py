"""ctypes binding of libloraine_hip.so (include/loraine_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or no GPU is
visible when a context is created, this raises."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libloraine_hip.so")

c_ctx = C.c_void_p
PD = C.POINTER(C.c_double)
PI64 = C.POINTER(C.c_int64)
PI = C.POINTER(C.c_int)
PPD = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); one entry per symbol declared in include/loraine_hip.h
SIGNATURES = {
    "lrn_create": (C.c_int, [C.POINTER(c_ctx), C.c_int]),
    "lrn_destroy": (C.c_int, [c_ctx]),
    "lrn_last_error": (C.c_char_p, [c_ctx]),
    "lrn_version": (C.c_int, []),
    "lrn_device_count": (C.c_int, []),
    "lrn_upload_model": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_void_p, PPD, PPD, PPD, PPD, PPD, PPD,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrn_synthetic_dense_model": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_uint64]),
    "lrn_synthetic_dense_problem": (C.c_int, [c_ctx, C.c_uint64, C.c_void_p, C.c_void_p, PD]),
    "lrn_get_constraint": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_void_p]),
    "lrn_set_option": (C.c_int, [c_ctx, C.c_char_p, C.c_double]),
    "lrn_set_shard": (C.c_int, [c_ctx, C.c_int, C.c_int]),
    "lrn_prepare_w": (C.c_int, [c_ctx, C.c_int] + [C.c_void_p] * 8 + [PI]),
    "lrn_set_scaling": (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_void_p]),
    "lrn_set_lin": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p]),
    "lrn_schur_assemble": (C.c_int, [c_ctx, C.c_int, C.c_void_p]),
    "lrn_schur_get": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_schur_add_diag": (C.c_int, [c_ctx, C.c_double]),
    "lrn_schur_factor": (C.c_int, [c_ctx, PI]),
    "lrn_schur_solve": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p]),
    "lrn_schur_shard_doubles": (C.c_int64, [c_ctx]),
    "lrn_schur_export_shard": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_schur_import_all": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_schur_is_partial_sum": (C.c_int, [c_ctx]),
    "lrn_schur_plan": (C.c_int, [c_ctx, C.c_int, PI]),
    "lrn_schur_export_full": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_schur_import_full": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_make_rhs": (C.c_int, [c_ctx, C.c_void_p, PPD, C.c_void_p]),
    "lrn_matvec": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p]),
    "lrn_matvec_partial": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p]),
    "lrn_prec_setup": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int, PI]),
    "lrn_prec_apply": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p]),
    "lrn_pcg": (C.c_int, [c_ctx, C.c_void_p, C.c_double, C.c_int, C.c_void_p, PI, PI]),
    "lrn_comm_unique_id": (C.c_int, [C.c_void_p]),
    "lrn_comm_init": (C.c_int, [c_ctx, C.c_void_p, C.c_int, C.c_int]),
    "lrn_comm_init_host": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrn_comm_destroy": (C.c_int, [c_ctx]),
    "lrn_comm_allreduce": (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.c_int]),
    "lrn_ip_set_c": (C.c_int, [c_ctx, C.c_int, C.c_void_p]),
    "lrn_ip_set_iterate": (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_void_p]),
    "lrn_ip_get_iterate": (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_void_p]),
    "lrn_ip_add_diag": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_double]),
    "lrn_ip_prepare_w": (C.c_int, [c_ctx, C.c_int, PI]),
    "lrn_ip_aa_x": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_ip_residual_d": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_ip_rhs_pred": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_ip_rhs_pred2": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p]),
    "lrn_ip_rhs_corr": (C.c_int, [c_ctx, C.c_double, C.c_void_p]),
    "lrn_ip_find_step": (C.c_int, [c_ctx, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrn_ip_update": (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrn_ip_stats": (C.c_int, [c_ctx, C.c_void_p]),
    "lrn_dbg_get_block": (C.c_int, [c_ctx, C.c_int, C.c_char_p, C.c_void_p, PI]),
    "lrn_dbg_tridiag_eig": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, PD, PI64]),
    "lrn_dbg_eigmin": (C.c_int, [c_ctx, C.c_int, C.c_void_p, PD, PI]),
    "lrn_dbg_lanczos": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, PD, PD, PI]),
    "lrn_get_timing": (C.c_int, [c_ctx, C.c_char_p, PD]),
    "lrn_get_count": (C.c_int64, [c_ctx, C.c_char_p]),
    "lrn_mfma_f64_peak": (C.c_int, [c_ctx, PD]),
    "lrn_hbm_copy_peak": (C.c_int, [c_ctx, C.c_int64, PD]),
    "lrn_xcc_probe": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "lrn_dbg_gemm": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                               C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "lrn_dbg_mfma_probe": (C.c_int, [c_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrn_dbg_potrf": (C.c_int, [c_ctx, C.c_int, C.c_void_p, PI]),
    "lrn_dbg_potrs": (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, PI]),
    "lrn_dbg_trsm": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, PI]),
    "lrn_dbg_svd_jacobi": (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, PI]),
}

GEMM_TRI_LOWER, GEMM_TRI_UPPER, GEMM_OFFDIAG_X2, GEMM_SQUARE, GEMM_KSEG_TRI, GEMM_SMALL_TILE = 1, 2, 4, 8, 16, 32

_lib = None


class LoraineHipError(RuntimeError):
    pass


def load_library():
    """Load libloraine_hip.so and bind every declared entry point. Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and shutil.which("hipcc") and os.environ.get("LRN_NO_AUTOBUILD") is None:
        # a fresh checkout on a box with the ROCm toolchain: compile the HIP sources once (this is the
        # product library itself, not a fallback)
        csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
        # one builder at a time: every rank of a torchrun job lands here together
        import fcntl
        with open(os.path.join(csrc, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not os.path.exists(LIB_PATH):
                print(f"[loraine.jl_amd] {LIB_PATH} missing: running `make -C {csrc}` (hipcc --offload-arch=gfx950)",
                      file=sys.stderr, flush=True)
                subprocess.run(["make", "-C", csrc, "-j8"], check=False, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise LoraineHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        # PyTorch-ROCm bundles its own HIP runtime; load it FIRST so that this library and
        # torch (device tensors, torch.distributed/RCCL) share one runtime in the process.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a):
    """Raw address of a numpy array / torch tensor / int address / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        # the library works on its own HIP stream: whatever torch still has queued for this tensor (its
        # zero-fill, an arithmetic result) must have landed before the pointer is handed over
        if getattr(a, "is_cuda", False):
            import torch
            torch.cuda.current_stream(a.device).synchronize()
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


def f64(a, order="F"):
    return np.require(a, dtype=np.float64, requirements=["F" if order == "F" else "C", "A"])


def ptr_array(arrs):
    """const T* const* from a list of numpy arrays (kept alive by the caller)."""
    arr = (C.c_void_p * max(1, len(arrs)))()
    for i, a in enumerate(arrs):
        arr[i] = a.ctypes.data if a is not None else None
    return arr
