"""Loader of the C-ABI library (include/cholmi.h).  Fails loudly when it is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LIBCHOLMI_PATH: another build of the same library, for same-box A/B measurements -- scripts/ only)
LIB_PATH = os.environ.get("LIBCHOLMI_PATH") or os.path.join(_HERE, "libcholmi.so")

_lib = None


class CholmiError(RuntimeError):
    """A libcholmi call returned a negative (argument / runtime) status."""

    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} failed with status {code}: {msg}")
        self.fn, self.code, self.msg = fn, code, msg


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dense_linear_app_amd/csrc`. There is no fallback path.")
    # torch ships its own HIP runtime under the same SONAME; importing it first makes
    # this process use ONE libamdhip64 for torch.distributed (RCCL) and for libcholmi.
    import torch  # noqa: F401

    L = C.CDLL(LIB_PATH)
    vp, i, d, u64 = C.c_void_p, C.c_int, C.c_double, C.c_ulonglong
    pp = C.POINTER(C.c_void_p)
    sig = {
        "chol_init": ([i, i], i),
        "chol_finalize": ([], i),
        "chol_set_device": ([i], i),
        "chol_last_error": ([], C.c_char_p),
        "chol_version": ([], C.c_char_p),
        "chol_set_rank": ([i, i], i),
        "chol_desc_create": ([pp, vp, i, i, i, i, i, i, i, i, i, i, i, i], i),
        "chol_desc_destroy": ([pp], i),
        "chol_potrf_tile": ([i, vp], i),
        "chol_trsm_tile": ([i, i, i, i, d, vp, vp], i),
        "chol_syrk_tile": ([i, i, d, vp, d, vp], i),
        "chol_gemm_tile": ([i, i, d, vp, vp, d, vp], i),
        "chol_plgsy_tile": ([d, i, vp, u64], i),
        "chol_lacpy_tile": ([i, vp, vp], i),
        "chol_lange_tile": ([i, vp, C.POINTER(d)], i),
        "chol_lauum_tile": ([i, vp], i),
        "chol_geadd_tile": ([i, d, vp, d, vp], i),
        "chol_potrs_tile": ([i, vp, vp], i),
        "chol_posv_tile": ([i, vp, vp], i),
        "chol_lapack_to_tile": ([vp, i, vp], i),
        "chol_tile_to_lapack": ([vp, vp, i], i),
        "chol_tile_upload": ([vp, i, i, vp], i),
        "chol_tile_download": ([vp, i, i, vp], i),
        "chol_residual_plgsy": ([vp, d, u64, C.POINTER(d)], i),
        "chol_residual_plgsy_inf": ([vp, d, u64, C.POINTER(d)], i),
        "chol_make_spd_like_chameleon": ([vp, i, i, d, C.c_char, u64], None),
        "chol_enforce_strict_diag_dominance": ([vp, i, i, d], None),
        "chol_extract_block": ([vp, i, i, i, i, i, vp], None),
        "chol_parse_payloads": ([C.c_char_p, vp, i, vp, vp, vp, vp], i),
        "chol_debug_flow_waves": ([], i),
        "chol_last_potrf_regimes": ([C.POINTER(i), C.POINTER(i)], i),
        "chol_debug_schedule_check": ([i, i, d, d, i, C.c_char_p, i], i),
        "chol_debug_schedule_check_grid": ([i, i, i, i, i, d, d, i, C.c_char_p, i], i),
        "chol_debug_comm_trace": ([i, i, i, i, i, d, d, C.c_char_p, i], i),
        "chol_debug_task_record": ([i, i], i),
        "chol_debug_task_check": ([C.POINTER(C.c_longlong)], i),
        "chol_last_potrf_stats": ([C.POINTER(d), C.POINTER(d), C.POINTER(i), C.POINTER(d)], i),
        "chol_set_profiling": ([i], i),
        "chol_debug_stamps": ([i, C.POINTER(C.c_ulonglong), i], i),
        "chol_mfma_probe": ([i, i, C.POINTER(d)], i),
        "chol_bench_update": ([vp, i, i, i, C.POINTER(d), C.POINTER(d)], i),
        "chol_desc_local_ptr": ([vp, C.POINTER(C.c_size_t)], vp),
        "chol_desc_local_tiles": ([vp, C.POINTER(i), C.POINTER(i)], i),
        "chol_set_transport": ([vp], i),
        "chol_transport_rccl_unique_id": ([vp], i),
        "chol_transport_rccl_init": ([vp, i, i], i),
        "chol_transport_rccl_finalize": ([], i),
        "chol_transport_rccl_version": ([], i),
        "chol_set_transport_channel": ([i, vp], i),
        "chol_transport_selftest": ([i, C.c_size_t], i),
        "chol_set_transport_null": ([], i),
        "chol_dist_rehearse": ([i, i, i, i, i, d, u64, vp, C.POINTER(d)], i),
        "chol_desc_set_version": ([vp, u64], i),
        "chol_tile_batch": ([i, i, i, i, vp, vp, vp, vp, vp, i], i),
        "chol_sync": ([], i),
        "chol_potrf_batch": ([i, i, i, vp, vp, vp, C.POINTER(i), i], i),
        "chol_batch_info": ([i, C.POINTER(i)], i),
        "chol_batch_stats": ([C.POINTER(C.c_longlong)], i),
        "chol_batch_mark": ([C.POINTER(C.c_ulonglong)], i),
        "chol_batch_wait": ([C.POINTER(C.c_ulonglong)], i),
        "chol_debug_calibration": ([C.POINTER(d)], i),
        "chol_debug_update_kernel": ([i, C.c_char_p, i], i),
        "chol_debug_device_counters": ([], i),
        "chol_dist_last_stats": ([C.POINTER(d), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)], i),
        "chol_dist_factorize_with": ([vp, vp, i, i, i, i, i, i], i),
        "chol_dist_gather_lower": ([vp, vp, i], i),
    }
    for name, (argt, rest) in sig.items():
        try:
            fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
        except AttributeError:
            if os.environ.get("LIBCHOLMI_PATH"):  # (an OLDER build for an A/B run may lack the newest diagnostics)
                continue
            raise
        fn.argtypes = argt
        fn.restype = rest
    _lib = L
    return L


ABI_SYMBOLS = None  # filled lazily by abi_symbols()


def abi_symbols() -> list[str]:
    """Every function include/cholmi.h declares (parsed from the header)."""
    import re

    hdr = os.path.join(_HERE, "..", "include", "cholmi.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chol_[a-z_0-9]+)\s*\(", text)))


def check(fn: str, code: int) -> int:
    """Negative statuses are programming/runtime errors -> raise; >= 0 is returned (LAPACK info)."""
    if code < 0:
        raise CholmiError(fn, code, lib().chol_last_error().decode(errors="replace"))
    return code
