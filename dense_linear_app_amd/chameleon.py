"""Chameleon-shaped Python surface over libcholmi.so.

Names follow the Chameleon C API the reference calls (chameleon.h is not in the
reference tree; call sites: worker_distrib.cpp:78, 238, 256, 323, 416, 511, 589 and
v6_test.c:41-56, 90-93), so reference-side code reads the same:

    CHAMELEON_Init(ncpu, ngpu)
    desc = CHAMELEON_Desc_Create(mat, ChamRealDouble, mb, nb, bsiz, lm, ln, i, j, m, n, p, q)
    info = CHAMELEON_dpotrf_Tile(ChamLower, desc)

`mat` may be None (library-owned HBM tile storage), a numpy array (host buffer, staged
around every call like the worker's blobs) or an int / torch tensor (device pointer,
tiles stay resident).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import CholmiError, check, lib

ChamRealFloat, ChamRealDouble = 2, 3
ChamNoTrans, ChamTrans = 111, 112
ChamUpper, ChamLower, ChamUpperLower = 121, 122, 123
ChamOneNorm, ChamFrobeniusNorm, ChamInfNorm, ChamMaxNorm = 171, 174, 175, 177
ChamNonUnit, ChamUnit = 131, 132
ChamLeft, ChamRight = 141, 142

_NP_OF = {ChamRealDouble: np.float64, ChamRealFloat: np.float32}


def CHAMELEON_Init(ncpu: int, ngpu: int) -> None:
    """W2:589 / V6:41.  Raises CholmiError(CHOL_ERR_NO_GPU) when no GPU is usable."""
    check("chol_init", lib().chol_init(int(ncpu), int(ngpu)))


def CHAMELEON_Finalize() -> None:
    """V6:93."""
    check("chol_finalize", lib().chol_finalize())


def set_device(device: int) -> None:
    check("chol_set_device", lib().chol_set_device(int(device)))


def set_rank(rank: int, nranks: int) -> None:
    check("chol_set_rank", lib().chol_set_rank(int(rank), int(nranks)))


def _ptr_of(mat):
    """-> (address or None, keepalive object)"""
    if mat is None:
        return None, None
    if isinstance(mat, np.ndarray):
        if not (mat.flags.f_contiguous or mat.flags.c_contiguous):
            raise ValueError("descriptor buffers must be contiguous")
        return mat.ctypes.data, mat
    if isinstance(mat, int):
        return mat, None
    if hasattr(mat, "data_ptr"):  # torch tensor
        return mat.data_ptr(), mat
    if isinstance(mat, (bytearray, memoryview)):
        buf = (C.c_char * len(mat)).from_buffer(mat)
        return C.addressof(buf), (mat, buf)
    raise TypeError(f"unsupported descriptor buffer type {type(mat)!r}")


class Desc:
    """CHAM_desc_t handle.  Destroyed explicitly (CHAMELEON_Desc_Destroy) or on GC."""

    def __init__(self, mat, dtype, mb, nb, bsiz, lm, ln, i, j, m, n, p, q):
        addr, self._keep = _ptr_of(mat)
        h = C.c_void_p()
        check("chol_desc_create",
              lib().chol_desc_create(C.byref(h), addr, dtype, mb, nb, bsiz, lm, ln, i, j, m, n, p, q))
        self._h = h
        self.dtype, self.mb, self.nb, self.bsiz = dtype, mb, nb, bsiz
        if (i, j, m, n) != (0, 0, lm, ln):
            # a sub-matrix view (library-owned storage): everything below works in view coordinates
            lm, ln = m, n
        self.lm, self.ln, self.m, self.n, self.p, self.q = lm, ln, m, n, p, q
        self.mt, self.nt = (lm + mb - 1) // mb, (ln + nb - 1) // nb

    @property
    def handle(self):
        if self._h is None:
            raise ValueError("descriptor already destroyed")
        return self._h

    @property
    def np_dtype(self):
        return _NP_OF[self.dtype]

    def destroy(self):
        if self._h is not None:
            h, self._h = self._h, None
            lib().chol_desc_destroy(C.byref(h))
            self._keep = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- resident-matrix helpers (extensions; Chameleon's Lapack_to_Tile / Tile_to_Lapack)
    def local_ptr(self) -> tuple[int, int]:
        n = C.c_size_t()
        p = lib().chol_desc_local_ptr(self.handle, C.byref(n))
        return int(p or 0), int(n.value)

    def set_version(self, version: int) -> None:
        """Name the content behind this 1-tile descriptor's device buffer (chol_desc_set_version)."""
        check("chol_desc_set_version", lib().chol_desc_set_version(self._h, C.c_ulonglong(int(version) & (2 ** 64 - 1))))

    def local_tiles(self) -> tuple[int, int]:
        a, b = C.c_int(), C.c_int()
        check("chol_desc_local_tiles", lib().chol_desc_local_tiles(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def from_lapack(self, A: np.ndarray) -> None:
        A = np.asfortranarray(A, dtype=self.np_dtype)
        check("chol_lapack_to_tile", lib().chol_lapack_to_tile(A.ctypes.data, A.shape[0], self.handle))

    def to_lapack(self) -> np.ndarray:
        A = np.zeros((self.lm, self.ln), dtype=self.np_dtype, order="F")
        check("chol_tile_to_lapack", lib().chol_tile_to_lapack(self.handle, A.ctypes.data, A.shape[0]))
        return A

    def upload_tile(self, I: int, J: int, tile: np.ndarray) -> None:
        t = np.asfortranarray(tile, dtype=self.np_dtype)
        assert t.size == self.bsiz
        check("chol_tile_upload", lib().chol_tile_upload(self.handle, I, J, t.ctypes.data))

    def download_tile(self, I: int, J: int) -> np.ndarray:
        t = np.empty((self.mb, self.nb), dtype=self.np_dtype, order="F")
        check("chol_tile_download", lib().chol_tile_download(self.handle, I, J, t.ctypes.data))
        return t


def CHAMELEON_Desc_Create(mat, dtype, mb, nb, bsiz, lm, ln, i, j, m, n, p, q) -> Desc:
    """W2:78, V6:44."""
    return Desc(mat, dtype, mb, nb, bsiz, lm, ln, i, j, m, n, p, q)


def CHAMELEON_Desc_Destroy(desc: Desc) -> None:
    """W2:256, V6:90."""
    desc.destroy()


def CHAMELEON_dpotrf_Tile(uplo: int, A: Desc) -> int:
    """W2:238, V6:56.  Returns info (0, or 1-based index of the first bad pivot)."""
    return check("chol_potrf_tile", lib().chol_potrf_tile(uplo, A.handle))


def CHAMELEON_dtrsm_Tile(side, uplo, trans, diag, alpha: float, A: Desc, B: Desc) -> int:
    """W2:323."""
    return check("chol_trsm_tile", lib().chol_trsm_tile(side, uplo, trans, diag, alpha, A.handle, B.handle))


def CHAMELEON_dsyrk_Tile(uplo, trans, alpha: float, A: Desc, beta: float, Cd: Desc) -> int:
    """W2:416."""
    return check("chol_syrk_tile", lib().chol_syrk_tile(uplo, trans, alpha, A.handle, beta, Cd.handle))


def CHAMELEON_dgemm_Tile(transA, transB, alpha: float, A: Desc, B: Desc, beta: float, Cd: Desc) -> int:
    """W2:511."""
    return check("chol_gemm_tile",
                 lib().chol_gemm_tile(transA, transB, alpha, A.handle, B.handle, beta, Cd.handle))


# single-precision names map to the same entry points (type comes from the descriptor)
CHAMELEON_spotrf_Tile = CHAMELEON_dpotrf_Tile
CHAMELEON_strsm_Tile = CHAMELEON_dtrsm_Tile
CHAMELEON_ssyrk_Tile = CHAMELEON_dsyrk_Tile
CHAMELEON_sgemm_Tile = CHAMELEON_dgemm_Tile


def CHAMELEON_dplgsy_Tile(bump: float, uplo: int, A: Desc, seed: int) -> int:
    """V6:46."""
    return check("chol_plgsy_tile", lib().chol_plgsy_tile(float(bump), uplo, A.handle, int(seed)))


CHAMELEON_splgsy_Tile = CHAMELEON_dplgsy_Tile


# -- the validation block of the reference driver (V6:51, 72-86)
def CHAMELEON_dlacpy_Tile(uplo: int, A: Desc, B: Desc) -> int:
    """V6:51, 79: B <- A on the `uplo` part (ChamUpperLower: everything)."""
    return check("chol_lacpy_tile", lib().chol_lacpy_tile(uplo, A.handle, B.handle))


def CHAMELEON_dlange_Tile(norm: int, A: Desc) -> float:
    """V6:74, 85.  Returns the norm, as Chameleon does."""
    r = C.c_double()
    check("chol_lange_tile", lib().chol_lange_tile(norm, A.handle, C.byref(r)))
    return r.value


def CHAMELEON_dlauum_Tile(uplo: int, A: Desc) -> int:
    """V6:80: tril(A) <- tril(L^T L) with L = tril(A)  (ChamLower)."""
    return check("chol_lauum_tile", lib().chol_lauum_tile(uplo, A.handle))


def CHAMELEON_dgeadd_Tile(trans: int, alpha: float, A: Desc, beta: float, B: Desc) -> int:
    """V6:83: B <- alpha A + beta B."""
    return check("chol_geadd_tile", lib().chol_geadd_tile(trans, float(alpha), A.handle, float(beta), B.handle))


def CHAMELEON_dpotrs_Tile(uplo: int, A: Desc, B: Desc) -> int:
    """B <- A^{-1} B with A = L L^T already factored by CHAMELEON_dpotrf_Tile (ChamLower)."""
    return check("chol_potrs_tile", lib().chol_potrs_tile(uplo, A.handle, B.handle))


def CHAMELEON_dposv_Tile(uplo: int, A: Desc, B: Desc) -> int:
    """Factor A and solve A X = B in place of B.  Returns info (> 0: A is not positive definite)."""
    return check("chol_posv_tile", lib().chol_posv_tile(uplo, A.handle, B.handle))


CHAMELEON_spotrs_Tile = CHAMELEON_dpotrs_Tile
CHAMELEON_sposv_Tile = CHAMELEON_dposv_Tile
CHAMELEON_slacpy_Tile = CHAMELEON_dlacpy_Tile
CHAMELEON_slange_Tile = CHAMELEON_dlange_Tile
CHAMELEON_slauum_Tile = CHAMELEON_dlauum_Tile
CHAMELEON_sgeadd_Tile = CHAMELEON_dgeadd_Tile


def residual_plgsy(L: Desc, bump: float, seed: int) -> float:
    """||tril(L)tril(L)^T - A||_F/||A||_F with A regenerated on the device (what V6:72-87 meant)."""
    r = C.c_double()
    check("chol_residual_plgsy", lib().chol_residual_plgsy(L.handle, float(bump), int(seed), C.byref(r)))
    return r.value


def residual_plgsy_inf(L: Desc, bump: float, seed: int) -> float:
    """||A - L L^T||_inf / ||A||_inf: the number V6:86 prints, computed correctly on the device."""
    r = C.c_double()
    check("chol_residual_plgsy_inf", lib().chol_residual_plgsy_inf(L.handle, float(bump), int(seed), C.byref(r)))
    return r.value


def last_potrf_stats() -> dict:
    t, u, f = C.c_double(), C.c_double(), C.c_double()
    n = C.c_int()
    lib().chol_last_potrf_stats(C.byref(t), C.byref(u), C.byref(n), C.byref(f))
    return {"total_ms": t.value, "update_ms": u.value, "update_launches": n.value, "update_flops": f.value}


REGIME_NAMES = ("paired", "plain", "halves", "counter_linked", "near_column", "flow", "yielding", "column_latency_form")


def last_potrf_regimes() -> dict:
    """How many waves of the last whole-matrix factorisation ran in which regime of the walker (chol_last_potrf_regimes):
    paired / plain / halves / near_column partition the waves that have an update; the others are attributes."""
    v, nt = (C.c_int * 8)(), C.c_int()
    lib().chol_last_potrf_regimes(v, C.byref(nt))
    d = {k: int(v[i]) for i, k in enumerate(REGIME_NAMES)}
    d["waves"] = int(nt.value)
    return d


def mfma_probe(dtype: int = ChamRealDouble, waves_per_simd: int = 1) -> float:
    """TFLOP/s of a register-only MFMA stream on every CU (sustained matrix-core ceiling)."""
    r = C.c_double()
    check("chol_mfma_probe", lib().chol_mfma_probe(dtype, waves_per_simd, C.byref(r)))
    return r.value


def bench_update(desc: Desc, k: int = 0, ablate: int = 0, reps: int = 3) -> tuple[float, float]:
    """(ms, TFLOP/s) of wave k's trailing-update launch alone (diagnostic; modifies the matrix)."""
    ms, fl = C.c_double(), C.c_double()
    check("chol_bench_update", lib().chol_bench_update(desc.handle, k, ablate, reps, C.byref(ms), C.byref(fl)))
    return ms.value, fl.value / (ms.value * 1e-3) / 1e12


def calibration() -> list:
    """[fp64 MFMA probe TFLOP/s, fp64 diagonal-block step us, fp32 ..., fp32 ...] + the four derived figures the
    walker's regime switches use (chol_debug_calibration)."""
    out = (C.c_double * 8)()
    check("chol_debug_calibration", lib().chol_debug_calibration(out))
    return list(out)


def update_kernel_name(dtype: int = ChamRealDouble) -> str:
    """Name (as rocprofv3 prints it) of the trailing-update kernel the library launches for `dtype` right now."""
    buf = C.create_string_buffer(96)
    check("chol_debug_update_kernel", lib().chol_debug_update_kernel(dtype, buf, len(buf)))
    return buf.value.decode()


def set_profiling(on: bool) -> None:
    lib().chol_set_profiling(1 if on else 0)


__all__ = [n for n in dir() if n.startswith(("CHAMELEON_", "Cham"))] + [
    "Desc", "CholmiError", "residual_plgsy", "residual_plgsy_inf", "last_potrf_stats", "set_profiling", "set_device", "set_rank"]
