"""GPU parity of the four tile operations (through the C ABI) against the CPU oracle.

Tolerance (fp64): per-op max|delta| <= 16*B*eps*max|ref|; fp32: same with
float eps.  Inputs: the reference generator's matrix (client_distrib.cpp:402-405) cut
into tiles, and the committed scipy/OpenBLAS fixture tests/golden/tileops_B64.npz.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
EPS = {np.float64: np.finfo(np.float64).eps, np.float32: np.finfo(np.float32).eps}


def tol(B, ref, dt=np.float64):
    return 16 * B * EPS[dt] * max(1.0, float(np.abs(ref).max()))


def desc1(ch, a):
    B = a.shape[0]
    dt = ch.ChamRealDouble if a.dtype == np.float64 else ch.ChamRealFloat
    return ch.CHAMELEON_Desc_Create(a, dt, B, B, B * B, B, B, 0, 0, B, B, 1, 1)


def tiles_for(orc, B, dt=np.float64):
    A = orc.reference_input(4 * B)
    g = lambda i, j: np.asfortranarray(orc.extract_block(A, B, i, j).astype(dt))
    return g(0, 0), g(1, 0), g(2, 0), g(1, 1), g(2, 1)


@pytest.mark.parametrize("B", [4, 12, 64, 128, 200, 256, 448, 512, 1024])
def test_tile_ops_fp64(cham, orc, B):
    ch = cham
    Akk, A10, A20, A11, A21 = tiles_for(orc, B)
    # POTRF (W2:238): lower triangle factored, strict upper untouched
    L = Akk.copy(order="F")
    d = desc1(ch, L)
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
    ch.CHAMELEON_Desc_Destroy(d)
    Lref, iref = orc.dpotrf(Akk)
    assert info == 0 and iref == 0
    assert np.abs(np.tril(L) - np.tril(Lref)).max() <= tol(B, Lref)
    assert np.array_equal(np.triu(L, 1), np.triu(Akk, 1))
    # TRSM (W2:323): reads only the lower triangle of L (its upper part is stale input)
    X10 = A10.copy(order="F")
    dl, da = desc1(ch, L), desc1(ch, X10)
    assert ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, dl, da) == 0
    X10ref = orc.dtrsm(Lref, A10)
    assert np.abs(X10 - X10ref).max() <= tol(B, X10ref)
    X20 = A20.copy(order="F")
    da2 = desc1(ch, X20)
    assert ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, dl, da2) == 0
    X20ref = orc.dtrsm(Lref, A20)
    assert np.abs(X20 - X20ref).max() <= tol(B, X20ref)
    # SYRK (W2:416): lower triangle updated, strict upper untouched
    C11 = A11.copy(order="F")
    dc = desc1(ch, C11)
    assert ch.CHAMELEON_dsyrk_Tile(ch.ChamLower, ch.ChamNoTrans, -1.0, da, 1.0, dc) == 0
    C11ref = orc.dsyrk(X10ref, A11)
    assert np.abs(np.tril(C11) - np.tril(C11ref)).max() <= tol(B, C11ref)
    assert np.array_equal(np.triu(C11, 1), np.triu(A11, 1))
    # GEMM (W2:511)
    C21 = A21.copy(order="F")
    dc2 = desc1(ch, C21)
    assert ch.CHAMELEON_dgemm_Tile(ch.ChamNoTrans, ch.ChamTrans, -1.0, da2, da, 1.0, dc2) == 0
    C21ref = orc.dgemm(X20ref, X10ref, A21)
    assert np.abs(C21 - C21ref).max() <= tol(B, C21ref)


def test_tile_ops_golden_fixture(cham):
    """Same flag sets against the committed scipy-OpenBLAS vectors (B = 64)."""
    ch = cham
    g = np.load(os.path.join(GOLD, "tileops_B64.npz"))
    B = 64
    L = np.asfortranarray(g["Akk"].copy())
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, L)) == 0
    assert np.abs(np.tril(L) - np.tril(g["potrf_out"])).max() <= tol(B, g["potrf_out"])
    X = np.asfortranarray(g["A10"].copy())
    ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, L), desc1(ch, X))
    assert np.abs(X - g["trsm10_out"]).max() <= tol(B, g["trsm10_out"])
    X2 = np.asfortranarray(g["A20"].copy())
    ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, L), desc1(ch, X2))
    assert np.abs(X2 - g["trsm20_out"]).max() <= tol(B, g["trsm20_out"])
    Cs = np.asfortranarray(g["A11"].copy())
    ch.CHAMELEON_dsyrk_Tile(ch.ChamLower, ch.ChamNoTrans, -1.0, desc1(ch, X), 1.0, desc1(ch, Cs))
    assert np.abs(np.tril(Cs) - np.tril(g["syrk11_out"])).max() <= tol(B, g["syrk11_out"])
    Cg = np.asfortranarray(g["A21"].copy())
    ch.CHAMELEON_dgemm_Tile(ch.ChamNoTrans, ch.ChamTrans, -1.0, desc1(ch, X2), desc1(ch, X), 1.0, desc1(ch, Cg))
    assert np.abs(Cg - g["gemm21_out"]).max() <= tol(B, g["gemm21_out"])


@pytest.mark.parametrize("B", [64, 256])
def test_tile_ops_fp32(cham, orc, B):
    ch = cham
    f = np.float32
    Akk, A10, A20, A11, A21 = tiles_for(orc, B, f)
    L = Akk.copy(order="F")
    assert ch.CHAMELEON_spotrf_Tile(ch.ChamLower, desc1(ch, L)) == 0
    Lref, _ = orc.spotrf(Akk)
    assert np.abs(np.tril(L) - np.tril(Lref)).max() <= tol(B, Lref, f)
    X = A10.copy(order="F")
    ch.CHAMELEON_strsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, L), desc1(ch, X))
    Xref = orc.strsm(Lref, A10)
    assert np.abs(X - Xref).max() <= tol(B, Xref, f)
    Cs = A11.copy(order="F")
    ch.CHAMELEON_ssyrk_Tile(ch.ChamLower, ch.ChamNoTrans, -1.0, desc1(ch, X), 1.0, desc1(ch, Cs))
    Csref = orc.ssyrk(Xref, A11)
    assert np.abs(np.tril(Cs) - np.tril(Csref)).max() <= tol(B, Csref, f)
    assert np.array_equal(np.triu(Cs, 1), np.triu(A11, 1))
    X2 = A20.copy(order="F")
    ch.CHAMELEON_strsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, L), desc1(ch, X2))
    Cg = A21.copy(order="F")
    ch.CHAMELEON_sgemm_Tile(ch.ChamNoTrans, ch.ChamTrans, -1.0, desc1(ch, X2), desc1(ch, X), 1.0, desc1(ch, Cg))
    Cgref = orc.sgemm(orc.strsm(Lref, A20), Xref, A21)
    assert np.abs(Cg - Cgref).max() <= tol(B, Cgref, f)


def test_alpha_beta_and_flags(cham, orc):
    """General alpha/beta; unsupported flag combinations are refused, not mis-computed."""
    ch = cham
    B = 128
    rng = np.random.default_rng(3)
    A = np.asfortranarray(rng.standard_normal((B, B)))
    Bm = np.asfortranarray(rng.standard_normal((B, B)))
    C0 = np.asfortranarray(rng.standard_normal((B, B)))
    Cg = C0.copy(order="F")
    ch.CHAMELEON_dgemm_Tile(ch.ChamNoTrans, ch.ChamTrans, 0.75, desc1(ch, A), desc1(ch, Bm), -0.5, desc1(ch, Cg))
    ref = orc.dgemm(A, Bm, C0, alpha=0.75, beta=-0.5)
    assert np.abs(Cg - ref).max() <= tol(B, ref)
    # beta = 0 must not read C (NaN in C must not propagate)
    Cn = np.full((B, B), np.nan, order="F")
    ch.CHAMELEON_dgemm_Tile(ch.ChamNoTrans, ch.ChamTrans, 1.0, desc1(ch, A), desc1(ch, Bm), 0.0, desc1(ch, Cn))
    assert np.abs(Cn - A @ Bm.T).max() <= tol(B, A @ Bm.T)
    Cs = C0.copy(order="F")
    ch.CHAMELEON_dsyrk_Tile(ch.ChamLower, ch.ChamNoTrans, 2.0, desc1(ch, A), 0.25, desc1(ch, Cs))
    ref = orc.dsyrk(A, C0, alpha=2.0, beta=0.25)
    assert np.abs(np.tril(Cs) - np.tril(ref)).max() <= tol(B, ref)
    Lm = np.asfortranarray(np.tril(rng.standard_normal((B, B))) + 8 * np.eye(B))
    X = C0.copy(order="F")
    ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, -1.5, desc1(ch, Lm), desc1(ch, X))
    ref = orc.dtrsm(Lm, C0, alpha=-1.5)
    assert np.abs(X - ref).max() <= tol(B, ref)
    with pytest.raises(ch.CholmiError):
        ch.CHAMELEON_dtrsm_Tile(ch.ChamLeft, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, Lm), desc1(ch, X))
    with pytest.raises(ch.CholmiError):
        ch.CHAMELEON_dgemm_Tile(ch.ChamTrans, ch.ChamTrans, 1.0, desc1(ch, A), desc1(ch, Bm), 0.0, desc1(ch, Cn))


def test_potrf_info_nonpositive_pivot(cham, orc):
    """LAPACK info semantics (W2:243): 1-based index of the first non-positive pivot."""
    ch = cham
    for B, bad in ((64, 10), (256, 200), (256, 0)):
        A = np.asfortranarray(orc.extract_block(orc.reference_input(B), B, 0, 0))
        A[bad, bad] = -1.0
        ref, iref = orc.dpotrf(A)
        got = A.copy(order="F")
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, got))
        assert info == iref == bad + 1
    # ragged client tile (C2:285,299-303): zero-padded diagonal block is singular
    A10 = orc.reference_input(10)
    t = np.asfortranarray(orc.extract_block(A10, 4, 2, 2))
    _, iref = orc.dpotrf(t)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, t.copy(order="F"))) == iref == 3


def test_device_resident_tile(cham, orc):
    """A descriptor may wrap a device pointer: the tile then stays in HBM (no staging)."""
    import torch

    ch = cham
    B = 256
    Akk = orc.extract_block(orc.reference_input(B), B, 0, 0)
    t = torch.from_numpy(np.ascontiguousarray(Akk.T)).cuda()  # column-major bytes
    d = ch.CHAMELEON_Desc_Create(t, ch.ChamRealDouble, B, B, B * B, B, B, 0, 0, B, B, 1, 1)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    L = t.cpu().numpy().T
    Lref, _ = orc.dpotrf(Akk)
    assert np.abs(np.tril(L) - np.tril(Lref)).max() <= tol(B, Lref)


@pytest.mark.parametrize("B", [12, 200, 256])
def test_potrf_upper_on_a_staged_host_tile(cham, orc, B):
    """ChamUpper on the worker's kind of tile (host buffer, any size): A = U^T U in the upper triangle,
    the strict lower triangle comes back untouched (v3 driver's --uplo U)."""
    ch = cham
    rng = np.random.default_rng(B)
    M = rng.uniform(-1, 1, (B, B))
    S = M @ M.T + B * np.eye(B)
    A = np.asfortranarray(np.triu(S) + np.tril(rng.uniform(-9, 9, (B, B)), -1))  # junk below the diagonal
    got = A.copy(order="F")
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamUpper, desc1(ch, got)) == 0
    Lref, info = orc.dpotrf(np.asfortranarray(S))
    assert info == 0
    assert np.abs(np.triu(got) - np.tril(Lref).T).max() <= tol(B, Lref)
    assert np.array_equal(np.tril(got, -1), np.tril(A, -1))
    bad = A.copy(order="F")
    bad[B // 2, B // 2] = -1.0
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamUpper, desc1(ch, bad)) == B // 2 + 1
