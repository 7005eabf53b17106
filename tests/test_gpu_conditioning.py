"""Parity where it is hard: ill-conditioned and badly scaled inputs, extreme pivots, NaN / Inf.

The reference's arithmetic is LAPACK dpotrf + BLAS dtrsm (W2:238, 323): a backward-stable
factorisation and a backward-stable substitution.  The GPU path solves with explicitly inverted
128 x 128 diagonal blocks, which alone is NOT backward stable (its residual grows with the
condition number of the block) -- so blocks whose inverse says kappa_inf(L11) is above a small
threshold get one step of iterative refinement inside the solve kernels.  These tests pin what
that buys, as a function of kappa (DESIGN.md section 6 carries the same table):

  * componentwise backward error, independent of kappa -- the bounds of Higham, Accuracy and
    Stability of Numerical Algorithms, Th. 10.3 (Cholesky) and Th. 8.5 (substitution), with the
    constant 8 B eps:   |L L^T - A| <= 8 B eps |L||L^T|,   |X L^T - A| <= 8 B eps (|X||L^T| + |A|);
  * forward error against the oracle (substitution TRSM, scalar-pivot POTRF) proportional to kappa:
    max|L - Lref| <= 16 B eps kappa_2(A) max|Lref|,  max|X - Xref| <= 16 B eps kappa_inf(L) max|Xref|
    (for graded matrices D A0 D, kappa of the equilibrated A0 and errors measured after unscaling:
    Cholesky and substitution are invariant under power-of-two diagonal scaling);
  * info identical to the oracle's for NaN / Inf / non-positive pivots; subnormal and huge pivots
    give the oracle's factor.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def desc1(ch, a):
    B = a.shape[0]
    return ch.CHAMELEON_Desc_Create(a, ch.ChamRealDouble, B, B, B * B, B, B, 0, 0, B, B, 1, 1)


def spd_spectral(n, kappa, seed):
    """Q diag(logspace(0, -log10 kappa)) Q^T: kappa_2 = kappa exactly (up to rounding)."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.logspace(0.0, -np.log10(kappa), n)
    A = (Q * lam) @ Q.T
    return np.asfortranarray((A + A.T) * 0.5)


def spd_graded(n, decades, seed):
    """D A0 D with A0 well conditioned and D = powers of two falling over `decades` decades: the
    leading blocks of L = D L0 have kappa up to 10^decades while the problem stays well posed.
    Returns (A, A0, d)."""
    rng = np.random.default_rng(seed)
    M = rng.uniform(-1, 1, (n, n))
    A0 = M @ M.T / n + np.eye(n)
    d = 2.0 ** (-np.round(np.linspace(0, decades * np.log2(10.0), n)))
    return np.asfortranarray(d[:, None] * A0 * d[None, :]), A0, d


def potrf_backward_ok(L, A, B):
    L = np.tril(L)
    E = np.abs(np.tril(L @ L.T - A))
    bound = 8 * B * EPS * (np.abs(L) @ np.abs(L).T)
    return bool((E <= bound + 1e-300).all()), float((E / (bound + 1e-300)).max())


def trsm_backward_ok(X, L, A, B):
    L = np.tril(L)
    E = np.abs(X @ L.T - A)
    bound = 8 * B * EPS * (np.abs(X) @ np.abs(L).T + np.abs(A))
    return bool((E <= bound + 1e-300).all()), float((E / (bound + 1e-300)).max())


@pytest.mark.parametrize("kappa", [1e2, 1e6, 1e10, 1e13])
@pytest.mark.parametrize("B", [128, 256, 512])
def test_tile_ops_prescribed_kappa(cham, orc, B, kappa):
    """POTRF -> TRSM -> SYRK/GEMM on Q diag(logspace) Q^T, against the oracle."""
    if kappa > 1e12 and B > 128:
        pytest.skip("kappa 1e13 is within n*eps of singular beyond one 128-block")
    ch = cham
    A = spd_spectral(2 * B, kappa, seed=int(np.log10(kappa)) * 1000 + B)
    Akk, A10, A11 = (np.asfortranarray(A[:B, :B]), np.asfortranarray(A[B:, :B]), np.asfortranarray(A[B:, B:]))
    kap = np.linalg.cond(Akk)
    L = Akk.copy(order="F")
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, L))
    Lref, iref = orc.dpotrf(Akk)
    assert info == iref == 0
    ok, worst = potrf_backward_ok(L, Akk, B)
    assert ok, f"POTRF componentwise backward error {worst:.2f} x bound"
    assert np.abs(np.tril(L) - np.tril(Lref)).max() <= 16 * B * EPS * kap * np.abs(Lref).max()
    assert np.array_equal(np.triu(L, 1), np.triu(Akk, 1))
    # TRSM against the factor the GPU itself produced (what the DAG does) and against the oracle's
    X = A10.copy(order="F")
    assert ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, L),
                                   desc1(ch, X)) == 0
    ok, worst = trsm_backward_ok(X, L, A10, B)
    assert ok, f"TRSM componentwise backward error {worst:.2f} x bound"
    Xref = orc.dtrsm(np.asfortranarray(np.tril(L)), A10)
    kl = np.linalg.cond(np.tril(L), np.inf)
    assert np.abs(X - Xref).max() <= 16 * B * EPS * kl * np.abs(Xref).max()
    # SYRK / GEMM with these operands: plain contractions, kappa does not enter
    C = A11.copy(order="F")
    assert ch.CHAMELEON_dsyrk_Tile(ch.ChamLower, ch.ChamNoTrans, -1.0, desc1(ch, X), 1.0, desc1(ch, C)) == 0
    Cref = orc.dsyrk(X, A11)
    scale = np.abs(X) @ np.abs(X).T + np.abs(A11)
    assert (np.abs(np.tril(C - Cref)) <= 16 * B * EPS * np.tril(scale)).all()
    G = A11.copy(order="F")
    assert ch.CHAMELEON_dgemm_Tile(ch.ChamNoTrans, ch.ChamTrans, -1.0, desc1(ch, X), desc1(ch, X), 1.0,
                                   desc1(ch, G)) == 0
    assert (np.abs(G - orc.dgemm(X, X, A11)) <= 16 * B * EPS * scale).all()


@pytest.mark.parametrize("decades", [6, 13, 20])
@pytest.mark.parametrize("B", [128, 512])
def test_tile_ops_graded(cham, orc, B, decades):
    """Graded D A0 D: kappa_inf of every leading 128-block of L is ~10^decades, the equilibrated
    problem is benign.  Errors are measured after unscaling."""
    ch = cham
    A, A0, d = spd_graded(2 * B, decades, seed=decades * 10 + B)
    Akk, A10 = np.asfortranarray(A[:B, :B]), np.asfortranarray(A[B:, :B])
    L = Akk.copy(order="F")
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, L))
    Lref, iref = orc.dpotrf(Akk)
    assert info == iref == 0
    assert np.linalg.cond(np.tril(L)[:128, :128], np.inf) > 10.0 ** (decades * 127 / (2 * B) - 1)
    ok, worst = potrf_backward_ok(L, Akk, B)
    assert ok, f"POTRF componentwise backward error {worst:.2f} x bound"
    k0 = np.linalg.cond(A0[:B, :B])
    dk = d[:B]
    assert np.abs((np.tril(L) - np.tril(Lref)) / dk[:, None]).max() <= 16 * B * EPS * k0 * np.abs(Lref / dk[:, None]).max()
    X = A10.copy(order="F")
    assert ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, L),
                                   desc1(ch, X)) == 0
    ok, worst = trsm_backward_ok(X, L, A10, B)
    assert ok, f"TRSM componentwise backward error {worst:.2f} x bound"
    Xref = orc.dtrsm(np.asfortranarray(np.tril(L)), A10)
    k0l = np.linalg.cond(np.tril(L) / dk[:, None], np.inf)
    rs = d[B:, None]
    assert np.abs((X - Xref) / rs).max() <= 16 * B * EPS * k0l * np.abs(Xref / rs).max()


@pytest.mark.parametrize("kl", [1e3, 1e8, 1e12])
def test_trsm_general_triangular_factor(cham, orc, kl):
    """TRSM is an ABI operation of its own (W2:323): L need not be a Cholesky factor.  The
    triangular factor of U diag(logspace(0, -log10 kl)) V^T: kappa_2(L) = kl, not graded."""
    ch = cham
    B = 256
    rng = np.random.default_rng(int(np.log10(kl)))
    U, _ = np.linalg.qr(rng.standard_normal((B, B)))
    V, _ = np.linalg.qr(rng.standard_normal((B, B)))
    _, R = np.linalg.qr(((U * np.logspace(0, -np.log10(kl), B)) @ V.T).T)
    Lm = np.asfortranarray(R.T + np.triu(np.full((B, B), np.nan), 1))  # strict upper: stale, must not be read
    A = np.asfortranarray(rng.uniform(-1, 1, (B, B)))
    X = A.copy(order="F")
    assert ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, Lm),
                                   desc1(ch, X)) == 0
    Lt = np.tril(np.nan_to_num(Lm))
    assert np.isfinite(X).all()
    ok, worst = trsm_backward_ok(X, Lt, A, B)
    assert ok, f"TRSM componentwise backward error {worst:.2f} x bound"
    Xref = orc.dtrsm(np.asfortranarray(Lt), A)
    assert np.abs(X - Xref).max() <= 16 * B * EPS * np.linalg.cond(Lt, np.inf) * np.abs(Xref).max()


@pytest.mark.parametrize("N,B,kappa", [(1024, 256, 1e2), (1024, 256, 1e6), (1024, 256, 1e10), (2048, 512, 1e8),
                                       (1536, 128, 1e9)])
def test_full_potrf_prescribed_kappa(cham, orc, N, B, kappa):
    """The whole wave DAG on an ill-conditioned matrix: residual independent of kappa, factor
    within 16 N eps kappa of the oracle's, same info."""
    ch = cham
    A = spd_spectral(N, kappa, seed=N + B)
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    d.from_lapack(A)
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
    Lref, iref = orc.cholesky_lower(A, B)
    assert info == iref == 0
    L = np.tril(d.to_lapack())
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-13
    assert potrf_backward_ok(L, A, N)[0]
    assert np.abs(L - Lref).max() <= 16 * N * EPS * kappa * np.abs(Lref).max()
    ores = np.linalg.norm(Lref @ Lref.T - A) / np.linalg.norm(A)
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= max(4 * ores, 1e-15)  # as good as substitution


@pytest.mark.parametrize("N,B,decades", [(1024, 256, 13), (2048, 512, 24)])
def test_full_potrf_graded(cham, orc, N, B, decades):
    ch = cham
    A, A0, dd = spd_graded(N, decades, seed=N)
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    d.from_lapack(A)
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
    Lref, iref = orc.cholesky_lower(A, B)
    assert info == iref == 0
    L = np.tril(d.to_lapack())
    ok, worst = potrf_backward_ok(L, A, N)
    assert ok, f"componentwise backward error {worst:.2f} x bound"
    assert np.abs((L - Lref) / dd[:, None]).max() <= 16 * N * EPS * np.linalg.cond(A0) * np.abs(Lref / dd[:, None]).max()


@pytest.mark.parametrize("B", [64, 128, 384])
@pytest.mark.parametrize("scale_exp", [-1010, -960, 960, 1010])
def test_pivots_near_the_ends_of_the_exponent_range(cham, orc, B, scale_exp):
    """The reference input scaled by 2^+-1010: pivots ~1e-302 / ~1e306 (DBL_MIN 2.2e-308, DBL_MAX
    1.8e308).  The scaling is exact, so the factor is the oracle's within the usual tolerance."""
    ch = cham
    A0 = orc.extract_block(orc.reference_input(B), B, 0, 0)
    A = np.asfortranarray(np.ldexp(A0, scale_exp))
    assert np.isfinite(A).all() and (np.diag(A) > 0).all()
    L = A.copy(order="F")
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, L))
    Lref, iref = orc.dpotrf(A)
    assert info == iref == 0
    assert np.isfinite(np.tril(L)).all()
    assert np.abs(np.tril(L) - np.tril(Lref)).max() <= 16 * B * EPS * np.abs(Lref).max()
    X = np.asfortranarray(np.ldexp(orc.extract_block(orc.reference_input(2 * B), B, 1, 0), scale_exp))
    X0 = X.copy(order="F")
    assert ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, desc1(ch, L),
                                   desc1(ch, X)) == 0
    Xref = orc.dtrsm(np.asfortranarray(np.tril(L)), X0)
    assert np.abs(X - Xref).max() <= 16 * B * EPS * np.abs(Xref).max()


def test_subnormal_and_infinite_pivots_follow_the_oracle(cham, orc):
    """A subnormal pivot is a positive pivot (LAPACK: sqrt, divide); +Inf passes `ajj > 0` too.
    Whatever the oracle's scalar code makes of them, the GPU must make the same."""
    ch = cham
    for B in (4, 64, 128):
        for j, v in ((0, 1e-310), (B // 2, 4e-320), (B - 1, 1e-315), (1, np.inf), (B - 1, np.inf)):
            A = np.asfortranarray(np.diag(np.linspace(1.0, 2.0, B)))
            A[j, j] = v
            if j + 1 < B:
                A[j + 1, j] = A[j, j + 1] = 0.0
            L = A.copy(order="F")
            info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, L))
            Lref, iref = orc.dpotrf(A)
            assert info == iref, (B, j, v, info, iref)
            a, b = np.tril(L), np.tril(Lref)
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b)), (B, j, v)
            m = np.isfinite(b)
            assert np.allclose(a[m], b[m], rtol=8 * EPS, atol=0.0), (B, j, v)
    # a subnormal pivot with a non-trivial column below it: L(i,j) = a(i,j) / sqrt(tiny) is large but finite
    B = 64
    A = np.asfortranarray(orc.extract_block(orc.reference_input(B), B, 0, 0))
    A[0, 0] = 3e-312
    A[1:, 0] = A[0, 1:] = np.linspace(-1.0, 1.0, B - 1) * 1e-157  # |a(i,0)| < sqrt(a(0,0) a(i,i))
    L = A.copy(order="F")
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, L))
    Lref, iref = orc.dpotrf(A)
    assert info == iref == 0
    assert np.abs(np.tril(L) - np.tril(Lref)).max() <= 16 * B * EPS * np.abs(Lref).max()
    assert abs(L[0, 0] - Lref[0, 0]) <= 4 * EPS * Lref[0, 0]


@pytest.mark.parametrize("B", [64, 256, 512])
def test_nan_input_gives_the_oracles_info(cham, orc, B):
    ch = cham
    base = np.asfortranarray(orc.extract_block(orc.reference_input(B), B, 0, 0))
    cases = [(5, 3), (B - 1, 0), (B // 2, B // 2), (B - 1, B - 2), (0, 0), (B // 2 + 3, B // 2 - 5)]
    for (i, j) in cases:
        A = base.copy(order="F")
        A[i, j] = np.nan
        got = A.copy(order="F")
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, got))
        _, iref = orc.dpotrf(A)
        assert info == iref and info > 0, (B, i, j, info, iref)
    # a NaN in the strict upper triangle is never read: info 0, same factor
    A = base.copy(order="F")
    A[3, 7] = np.nan
    got = A.copy(order="F")
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, desc1(ch, got)) == 0
    Lref, _ = orc.dpotrf(base)
    assert np.abs(np.tril(got) - np.tril(Lref)).max() <= 16 * B * EPS * np.abs(Lref).max()
    assert np.isnan(got[3, 7])


def test_nan_in_a_panel_tile_whole_matrix(cham, orc):
    """NaN below the diagonal tile: the TRSM spreads it over its row, the SYRK onto the diagonal of a
    later tile; info is the oracle's (first NaN pivot, global index)."""
    ch = cham
    N, B = 1024, 256
    base = orc.reference_input(N)
    for (i, j) in ((700, 100), (300, 299), (1023, 0), (600, 520)):
        A = base.copy(order="F")
        A[i, j] = A[j, i] = np.nan
        d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
        d.from_lapack(A)
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        _, iref = orc.cholesky_lower(A, B)
        assert info == iref and info > 0, (i, j, info, iref)
