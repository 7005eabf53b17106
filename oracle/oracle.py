"""ctypes binding of oracle/libchol_oracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (dense_linear_app_amd) must never import this module.
See the header of chol_oracle.c for what is and is not pinned to the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libchol_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libref_client.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="F_CONTIGUOUS")
_dpc = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")  # 1-D buffers
_fp = np.ctypeslib.ndpointer(dtype=np.float32, flags="F_CONTIGUOUS")
_fpc = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc).  Returns the .so path."""
    src = os.path.join(_HERE, "chol_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libchol_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def build_ref() -> str | None:
    """Compile the reference's own input functions (only where /root/reference exists)."""
    if os.path.isdir(os.environ.get("REFERENCE_ROOT", "/root/reference")):
        subprocess.check_call(["sh", os.path.join(_HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)
    return _REF_SO if os.path.exists(_REF_SO) else None


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_mt64_draws.argtypes = [C.c_uint64, C.c_int, np.ctypeslib.ndpointer(np.uint64)]
        L.orc_make_spd_like_chameleon.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_char, C.c_uint64]
        L.orc_enforce_strict_diag_dominance.argtypes = [_dp, C.c_int, C.c_int, C.c_double]
        L.orc_extract_block.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.orc_lcg_jump.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_lcg_jump.restype = C.c_uint64
        L.orc_plgsy_entry.argtypes = [C.c_double, C.c_uint64, C.c_int64, C.c_int64, C.c_int64]
        L.orc_plgsy_entry.restype = C.c_double
        L.orc_plgsy_matrix.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_uint64]
        L.orc_plgsy_tiles.argtypes = [_dpc, C.c_int, C.c_int, C.c_double, C.c_uint64]
        L.orc_plgsy_tiles_lower.argtypes = [_dpc, C.c_int, C.c_int, C.c_double, C.c_uint64]
        L.orc_plgsy_tiles_lower_of.argtypes = [_dpc, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_int64]
        L.orc_plgsy_tiles_lower_of.restype = None
        L.orc_plgsy_tiles_lower_at.argtypes = [_dpc, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_int64, C.c_int]
        L.orc_plgsy_tiles_lower_at.restype = None
        L.orc_dgemm_nt.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, C.c_int,
                                   C.c_double, _dp, C.c_int]
        L.orc_dsyrk_ln.argtypes = [C.c_int, C.c_int, C.c_double, _dp, C.c_int, C.c_double, _dp, C.c_int]
        L.orc_dtrsm_rltn.argtypes = [C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, C.c_int]
        L.orc_dpotrf_lower.argtypes = [C.c_int, _dp, C.c_int]
        L.orc_dpotrf_lower.restype = C.c_int
        L.orc_tiled_potrf_lower.argtypes = [_dpc, C.c_int, C.c_int, C.c_int]
        L.orc_tiled_potrf_lower.restype = C.c_int
        L.orc_num_threads.restype = C.c_int
        L.orc_lapack_to_tile.argtypes = [_dp, C.c_int, C.c_int, C.c_int, _dpc]
        L.orc_tile_to_lapack.argtypes = [_dpc, C.c_int, C.c_int, C.c_int, _dp]
        L.orc_residual_lower.argtypes = [_dp, _dp, C.c_int, C.c_int]
        L.orc_residual_lower.restype = C.c_double
        L.orc_sgemm_nt.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, _fp, C.c_int, _fp, C.c_int,
                                   C.c_float, _fp, C.c_int]
        L.orc_ssyrk_ln.argtypes = [C.c_int, C.c_int, C.c_float, _fp, C.c_int, C.c_float, _fp, C.c_int]
        L.orc_strsm_rltn.argtypes = [C.c_int, C.c_int, C.c_float, _fp, C.c_int, _fp, C.c_int]
        L.orc_spotrf_lower.argtypes = [C.c_int, _fp, C.c_int]
        L.orc_spotrf_lower.restype = C.c_int
        L.orc_tiled_spotrf_lower.argtypes = [_fpc, C.c_int, C.c_int, C.c_int]
        L.orc_tiled_spotrf_lower.restype = C.c_int
        _lib = L
    return _lib


# --------------------------------------------------------------------------- inputs
def make_spd_like_chameleon(N: int, bump: float = 100.0, uplo: str = "L", seed: int = 12345) -> np.ndarray:
    """C2:224-252 (called at C2:404 with bump=100, 'L', seed=12345)."""
    A = np.zeros((N, N), dtype=np.float64, order="F")
    lib().orc_make_spd_like_chameleon(A, N, N, bump, uplo.encode()[:1], seed)
    return A


def enforce_strict_diag_dominance(A: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """C2:255-264, in place."""
    N = A.shape[0]
    lib().orc_enforce_strict_diag_dominance(A, N, A.shape[0], eps)
    return A


def reference_input(N: int) -> np.ndarray:
    """The matrix the reference client factors (C2:402-405)."""
    return enforce_strict_diag_dominance(make_spd_like_chameleon(N))


def extract_block(A: np.ndarray, B: int, bi: int, bj: int) -> np.ndarray:
    """C2:280-309: zero-padded B x B column-major tile."""
    N = A.shape[0]
    blk = np.zeros((B, B), dtype=np.float64, order="F")
    lib().orc_extract_block(np.asfortranarray(A), N, N, B, bi, bj, blk)
    return blk


def lcg_jump(n: int, seed: int) -> int:
    """State of Chameleon's plgsy LCG n steps after `seed` (published core_dplgsy jump-ahead)."""
    return int(lib().orc_lcg_jump(n, seed))


def plgsy_entry(bump: float, seed: int, N: int, i: int, j: int) -> float:
    """Entry (i, j) of the order-N matrix CHAMELEON_dplgsy_Tile(bump, ., ., seed) generates (V6:46)."""
    return lib().orc_plgsy_entry(bump, seed, N, i, j)


def plgsy_matrix(N: int, bump: float, seed: int) -> np.ndarray:
    """The full symmetric matrix, LAPACK layout."""
    A = np.empty((N, N), dtype=np.float64, order="F")
    lib().orc_plgsy_matrix(A, N, N, bump, seed)
    return A


def plgsy_tiles(Nb: int, B: int, bump: float, seed: int) -> np.ndarray:
    """Tile-layout matrix (Nb*Nb tiles of B*B doubles, tile order column-major)."""
    T = np.empty(Nb * Nb * B * B, dtype=np.float64)
    lib().orc_plgsy_tiles(T, Nb, B, bump, seed)
    return T


# --------------------------------------------------------------------------- tile ops
def plgsy_tiles_lower(Nb: int, B: int, bump: float, seed: int, order: int | None = None) -> np.ndarray:
    """Tile layout with only the tiles on or below the diagonal filled (the others are zero): the
    factorisation's input, generated one tile column at a time (fast).  order > Nb*B: the leading Nb x Nb
    tiles of the larger matrix of that order."""
    T = np.zeros(Nb * Nb * B * B, dtype=np.float64)
    lib().orc_plgsy_tiles_lower_of(T, Nb, B, float(bump), int(seed), int(order if order else Nb * B))
    return T


def plgsy_tiles_lower_at(Nb: int, B: int, bump: float, seed: int, order: int, first: int) -> np.ndarray:
    """The Nb x Nb lower tiles of the diagonal block that starts at tile (first, first) of the order-`order` matrix."""
    T = np.zeros(Nb * Nb * B * B, dtype=np.float64)
    lib().orc_plgsy_tiles_lower_at(T, Nb, B, float(bump), int(seed), int(order), int(first))
    return T


def dpotrf(A: np.ndarray) -> tuple[np.ndarray, int]:
    """CHAMELEON_dpotrf_Tile(ChamLower) on one tile (W2:238). Returns (tile, info)."""
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    info = lib().orc_dpotrf_lower(A.shape[0], A, A.shape[0])
    return A, info


def dtrsm(L: np.ndarray, A: np.ndarray, alpha: float = 1.0) -> np.ndarray:
    """CHAMELEON_dtrsm_Tile(Right, Lower, Trans, NonUnit, alpha, L, A) (W2:323)."""
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    L = np.asfortranarray(L, dtype=np.float64)
    lib().orc_dtrsm_rltn(A.shape[0], A.shape[1], alpha, L, L.shape[0], A, A.shape[0])
    return A


def dsyrk(A: np.ndarray, Cm: np.ndarray, alpha: float = -1.0, beta: float = 1.0) -> np.ndarray:
    """CHAMELEON_dsyrk_Tile(Lower, NoTrans, alpha, A, beta, C) (W2:416)."""
    Cm = np.array(Cm, dtype=np.float64, order="F", copy=True)
    A = np.asfortranarray(A, dtype=np.float64)
    lib().orc_dsyrk_ln(Cm.shape[0], A.shape[1], alpha, A, A.shape[0], beta, Cm, Cm.shape[0])
    return Cm


def dgemm(Ai: np.ndarray, Aj: np.ndarray, Cm: np.ndarray, alpha: float = -1.0, beta: float = 1.0) -> np.ndarray:
    """CHAMELEON_dgemm_Tile(NoTrans, Trans, alpha, Ai, Aj, beta, C) (W2:511)."""
    Cm = np.array(Cm, dtype=np.float64, order="F", copy=True)
    Ai = np.asfortranarray(Ai, dtype=np.float64)
    Aj = np.asfortranarray(Aj, dtype=np.float64)
    lib().orc_dgemm_nt(Cm.shape[0], Cm.shape[1], Ai.shape[1], alpha, Ai, Ai.shape[0], Aj, Aj.shape[0],
                       beta, Cm, Cm.shape[0])
    return Cm


def spotrf(A):
    A = np.array(A, dtype=np.float32, order="F", copy=True)
    info = lib().orc_spotrf_lower(A.shape[0], A, A.shape[0])
    return A, info


def strsm(L, A, alpha=1.0):
    A = np.array(A, dtype=np.float32, order="F", copy=True)
    L = np.asfortranarray(L, dtype=np.float32)
    lib().orc_strsm_rltn(A.shape[0], A.shape[1], alpha, L, L.shape[0], A, A.shape[0])
    return A


def ssyrk(A, Cm, alpha=-1.0, beta=1.0):
    Cm = np.array(Cm, dtype=np.float32, order="F", copy=True)
    A = np.asfortranarray(A, dtype=np.float32)
    lib().orc_ssyrk_ln(Cm.shape[0], A.shape[1], alpha, A, A.shape[0], beta, Cm, Cm.shape[0])
    return Cm


def sgemm(Ai, Aj, Cm, alpha=-1.0, beta=1.0):
    Cm = np.array(Cm, dtype=np.float32, order="F", copy=True)
    Ai = np.asfortranarray(Ai, dtype=np.float32)
    Aj = np.asfortranarray(Aj, dtype=np.float32)
    lib().orc_sgemm_nt(Cm.shape[0], Cm.shape[1], Ai.shape[1], alpha, Ai, Ai.shape[0], Aj, Aj.shape[0],
                       beta, Cm, Cm.shape[0])
    return Cm


# --------------------------------------------------------------------------- whole matrix
def lapack_to_tile(A: np.ndarray, B: int) -> np.ndarray:
    N = A.shape[0]
    Nb = (N + B - 1) // B
    T = np.empty(Nb * Nb * B * B, dtype=np.float64)
    lib().orc_lapack_to_tile(np.asfortranarray(A), N, N, B, T)
    return T


def tile_to_lapack(T: np.ndarray, N: int, B: int) -> np.ndarray:
    A = np.zeros((N, N), dtype=np.float64, order="F")
    lib().orc_tile_to_lapack(np.ascontiguousarray(T), N, N, B, A)
    return A


def tiled_potrf(T: np.ndarray, Nb: int, B: int, nthreads: int = 0) -> int:
    """The reference wave DAG (C2:506-565) on a tile-layout matrix, in place."""
    if T.dtype == np.float32:
        return lib().orc_tiled_spotrf_lower(T, Nb, B, nthreads)
    return lib().orc_tiled_potrf_lower(T, Nb, B, nthreads)


def cholesky_lower(A: np.ndarray, B: int, nthreads: int = 0) -> tuple[np.ndarray, int]:
    """Factor a LAPACK-layout SPD matrix through the tile DAG; returns (tril(L), info)."""
    N = A.shape[0]
    assert N % B == 0
    T = lapack_to_tile(A, B)
    info = tiled_potrf(T, N // B, B, nthreads)
    return np.tril(tile_to_lapack(T, N, B)), info


def residual_lower(L: np.ndarray, A: np.ndarray) -> float:
    """||tril(L) tril(L)^T - A||_F / ||A||_F (the check V6:72-87 intended)."""
    N = A.shape[0]
    return lib().orc_residual_lower(np.asfortranarray(L), np.asfortranarray(A), N, N)


def num_threads() -> int:
    return lib().orc_num_threads()


# --------------------------------------------------------------------------- V6 validation sequence
def cholesky_lower_any(A: np.ndarray, B: int, nthreads: int = 0) -> tuple[np.ndarray, int]:
    """cholesky_lower for any N: the matrix is extended by an identity block to a multiple of B
    (which leaves the factor of the leading N x N part unchanged) and cropped afterwards."""
    N = A.shape[0]
    Np = (N + B - 1) // B * B
    if Np == N:
        return cholesky_lower(A, B, nthreads)
    Ap = np.eye(Np, dtype=np.float64, order="F")
    Ap[:N, :N] = A
    L, info = cholesky_lower(Ap, B, nthreads)
    return np.asfortranarray(L[:N, :N]), (info if info <= N else 0)


def cham_plgsy_visible(N: int, NB: int, bump: float, seed: int, uplo: str = "L") -> np.ndarray:
    """What CHAMELEON_dplgsy_Tile(bump, uplo, desc, seed) leaves in a zero-initialised N x N
    descriptor with NB x NB tiles (V6:44-46): the tiles on the `uplo` side, and the diagonal
    tiles in full (both triangles); the tiles on the other side are not touched."""
    full = plgsy_matrix(N, bump, seed)
    if uplo == "A":
        return full
    vis = np.tril(full) if uplo == "L" else np.triu(full)
    for t in range(0, N, NB):
        e = min(N, t + NB)
        vis[t:e, t:e] = full[t:e, t:e]
    return np.asfortranarray(vis)


def cham_lange_inf(A: np.ndarray) -> float:
    """CHAMELEON_dlange_Tile(ChamInfNorm, A) (V6:74, 85): max row sum of |a_ij|."""
    return float(np.abs(A).sum(axis=1).max())


def cham_lauum_lower(A: np.ndarray) -> np.ndarray:
    """CHAMELEON_dlauum_Tile(ChamLower, A) (V6:80): tril(A) <- tril(L^T L), L = tril(A);
    the strict upper triangle is not referenced (LAPACK dlauum)."""
    L = np.tril(A)
    out = np.array(A, order="F", copy=True)
    il = np.tril_indices(A.shape[0])
    out[il] = (L.T @ L)[il]
    return out


def v6_literal_validation(N: int, NB: int, seed: int = 42, nthreads: int = 0) -> dict:
    """The validation sequence of v6_test.c exactly as written (V6:44-86), which is what produced
    column 12 (rel_error) of the reference's bench.csv:

        A     <- dplgsy(bump=N, ChamLower, seed)           lower tiles + full diagonal tiles
        Aorig <- dlacpy(ChamUpperLower, A)
        A     <- dpotrf(ChamLower, A)
        normA <- dlange(Inf, Aorig)
        R     <- 0; dlacpy(ChamLower, A -> R); dlauum(ChamLower, R)     R = tril(L^T L), not L L^T
        Aorig <- Aorig - R   (dgeadd)                       over the whole matrix
        rel   <- dlange(Inf, Aorig) / normA

    It is not a residual of the factorisation: it is dominated by the strict
    upper triangles of the diagonal tiles, which R does not have.  Restated because the recorded
    values are the only numerical outputs of the reference and pin the generator and this
    sequence to three digits."""
    full = plgsy_matrix(N, float(N), seed)
    vis = cham_plgsy_visible(N, NB, float(N), seed, "L")
    L, info = cholesky_lower_any(full, NB, nthreads)
    normA = cham_lange_inf(vis)
    R = np.zeros((N, N), order="F")
    il = np.tril_indices(N)
    R[il] = L[il]
    R = cham_lauum_lower(R)
    D = vis - R
    res = cham_lange_inf(D)
    return {"info": info, "normA": normA, "residual": res, "rel": res / (normA if normA > 0 else 1.0)}


# --------------------------------------------------------------------------- reference build
class RefClient:
    """The reference's own functions (oracle/_ref, built by build_ref.sh)."""

    def __init__(self):
        path = build_ref()
        if path is None:
            raise FileNotFoundError("oracle/_ref/libref_client.so (needs /root/reference)")
        L = C.CDLL(path)
        L.ref_make_spd_like_chameleon.argtypes = [_dp, C.c_int, C.c_int, C.c_double, C.c_char, C.c_uint64]
        L.ref_enforce_strict_diag_dominance.argtypes = [_dp, C.c_int, C.c_int]
        L.ref_extract_block.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.ref_block_id_from_ij.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.ref_parse_int_str.argtypes = [C.c_char_p, C.c_int]
        L.ref_load_params.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.L = L

    def reference_input(self, N: int, bump=100.0, uplo="L", seed=12345, dominance=True) -> np.ndarray:
        A = np.zeros((N, N), dtype=np.float64, order="F")
        self.L.ref_make_spd_like_chameleon(A, N, N, bump, uplo.encode()[:1], seed)
        if dominance:
            self.L.ref_enforce_strict_diag_dominance(A, N, N)
        return A

    def extract_block(self, A, B, bi, bj):
        blk = np.zeros((B, B), dtype=np.float64, order="F")
        self.L.ref_extract_block(np.asfortranarray(A), A.shape[0], A.shape[0], B, bi, bj, blk)
        return blk

    def block_id_from_ij(self, i, j) -> str:
        buf = C.create_string_buffer(64)
        self.L.ref_block_id_from_ij(i, j, buf, 64)
        return buf.value.decode()

    def parse_int_str(self, s: str, fallback: int) -> int:
        return self.L.ref_parse_int_str(s.encode(), fallback)

    def load_params(self, argv: list[str]) -> tuple[int, int]:
        """argv excludes the program name.  Reads CHOLESKY_N/B from this process's env."""
        full = [b"app"] + [a.encode() for a in argv]
        arr = (C.c_char_p * len(full))(*full)
        n, b = C.c_int(), C.c_int()
        rc = self.L.ref_load_params(len(full), arr, C.byref(n), C.byref(b))
        if rc:
            raise ValueError("load_params threw")
        return n.value, b.value
