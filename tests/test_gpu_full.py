"""GPU parity of the whole-matrix tiled POTRF (driver path, v6_test.c:44-56) against the
CPU oracle's wave DAG on the same input, plus size-independent properties at scale.

Stated tolerances: fp64 ||tril(L)tril(L)^T - A||_F/||A||_F <= 1e-13 and
GPU-vs-oracle max|dL|/max|L| <= 1e-12; fp32 residual <= 5e-5, max|dL|/max|L| <= 1e-4.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def full_desc(ch, N, B, dtype=None):
    dtype = dtype or ch.ChamRealDouble
    return ch.CHAMELEON_Desc_Create(None, dtype, B, B, B * B, N, N, 0, 0, N, N, 1, 1)


@pytest.mark.parametrize("N,B", [(1024, 256), (1024, 128), (2048, 512), (3072, 1024)])
def test_full_potrf_reference_input(cham, orc, N, B):
    ch = cham
    A = orc.reference_input(N)
    d = full_desc(ch, N, B)
    d.from_lapack(A)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    L = np.tril(d.to_lapack())
    Lref, info = orc.cholesky_lower(A, B)
    assert info == 0
    assert np.abs(L - Lref).max() / np.abs(Lref).max() <= 1e-12
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-13
    # strict upper tiles and the strict upper triangle of diagonal tiles are untouched
    assert np.array_equal(np.triu(d.to_lapack(), 1), np.triu(A, 1))


def test_full_potrf_golden_config1(cham):
    """BASELINE config 1 (N=1024, B=256) against the committed scipy-replay fixture."""
    from oracle import oracle as orc

    ch = cham
    g = np.load(os.path.join(GOLD, "dag_N1024_B256.npz"))
    A = orc.reference_input(1024)
    d = full_desc(ch, 1024, 256)
    d.from_lapack(A)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    L = np.tril(d.to_lapack())
    scale = np.abs(g["diag"]).max()
    assert np.abs(np.diag(L) - g["diag"]).max() / scale <= 1e-12
    assert np.abs(L[g["probe_i"], g["probe_j"]] - g["probe_v"]).max() / scale <= 1e-12
    fro = np.array([[np.linalg.norm(L[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256]) for j in range(4)] for i in range(4)])
    assert np.abs(fro - g["tile_fro"]).max() <= 1e-10


def test_plgsy_matches_oracle_bits(cham, orc):
    ch = cham
    N, B = 1024, 256
    d = full_desc(ch, N, B)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamUpperLower, d, 42)
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    A = orc.tile_to_lapack(T, N, B)
    assert np.array_equal(d.to_lapack().view(np.uint64), A.view(np.uint64))
    assert np.array_equal(A, A.T)
    # one-sided generation (Chameleon's rule): the tiles on that side + the diagonal tiles in
    # full; the other tiles keep what they held (zeros in library-allocated storage)
    for uplo, side in ((ch.ChamLower, "L"), (ch.ChamUpper, "U")):
        for n, b in ((1024, 256), (1000, 192)):
            e = full_desc(ch, n, b)
            ch.CHAMELEON_dplgsy_Tile(float(n), uplo, e, 42)
            assert np.array_equal(e.to_lapack(), orc.cham_plgsy_visible(n, b, float(n), 42, side))


@pytest.mark.parametrize("N,B", [(4096, 512), (8192, 1024)])
def test_full_potrf_plgsy_vs_oracle(cham, orc, N, B):
    ch = cham
    d = full_desc(ch, N, B)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    res = ch.residual_plgsy(d, float(N), 42)
    assert res <= 1e-13
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = np.tril(orc.tile_to_lapack(T, N, B))
    L = np.tril(d.to_lapack())
    assert np.abs(L - Lref).max() / np.abs(Lref).max() <= 1e-12


def test_residual_kernel_detects_errors(cham):
    """The on-device residual is a real check: a corrupted factor must fail it."""
    ch = cham
    N, B = 2048, 512
    d = full_desc(ch, N, B)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 7)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    assert ch.residual_plgsy(d, float(N), 7) <= 1e-13
    t = d.download_tile(2, 1)
    t[5, 7] += 1e-3
    d.upload_tile(2, 1, t)
    assert ch.residual_plgsy(d, float(N), 7) > 1e-9


def test_full_potrf_info(cham, orc):
    """Non-SPD input: info = 1-based GLOBAL index of the failing pivot (LAPACK / W2:243)."""
    ch = cham
    N, B = 1024, 256
    A = orc.reference_input(N)
    A[700, 700] = -5.0
    d = full_desc(ch, N, B)
    d.from_lapack(A)
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
    _, iref = orc.cholesky_lower(A, B)
    assert info == iref == 701


def test_full_potrf_fp32(cham, orc):
    ch = cham
    N, B = 2048, 512
    d = full_desc(ch, N, B, ch.ChamRealFloat)
    ch.CHAMELEON_splgsy_Tile(float(N), ch.ChamLower, d, 42)
    assert ch.CHAMELEON_spotrf_Tile(ch.ChamLower, d) == 0
    assert ch.residual_plgsy(d, float(N), 42) <= 5e-5
    T = orc.plgsy_tiles(N // B, B, float(N), 42).astype(np.float32)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = np.tril(orc.tile_to_lapack(T.astype(np.float64), N, B))
    L = np.tril(d.to_lapack().astype(np.float64))
    assert np.abs(L - Lref).max() / np.abs(Lref).max() <= 1e-4


@pytest.mark.parametrize("N,B", [(16384, 512), (32768, 512), (32768, 1024)])
def test_full_potrf_large_properties(cham, N, B):
    """BASELINE configs 2 (16384/512) and 3 (32768/512) and the headline's smaller sibling:
    residual of the regenerated matrix, idempotent regeneration, positivity."""
    ch = cham
    d = full_desc(ch, N, B)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    assert ch.residual_plgsy(d, float(N), 42) <= 1e-13
    for k in (0, N // B - 1):
        t = d.download_tile(k, k)
        assert np.isfinite(np.tril(t)).all() and np.diag(t).min() > 0


def _compare_leading_tiles(ch, orc, d, N, B, Nb_lead, tol):
    """The leading Nb_lead x Nb_lead tiles of the factor in `d` against the oracle's factorisation of the leading
    block of the same matrix (the leading block of a Cholesky factor is the factor of the leading block), tile by tile."""
    T = orc.plgsy_tiles_lower(Nb_lead, B, float(N), 42, order=N)
    assert orc.tiled_potrf(T, Nb_lead, B) == 0
    T = T.reshape(Nb_lead * Nb_lead, B, B)
    scale = max(np.abs(T[I + J * Nb_lead]).max() for J in range(Nb_lead) for I in range(J, Nb_lead))
    worst = 0.0
    for J in range(Nb_lead):
        for I in range(J, Nb_lead):
            ref = T[I + J * Nb_lead].T  # (tile stored column-major: element (ii, jj) at ii + jj*B)
            got = d.download_tile(I, J)
            diff = np.tril(got) - np.tril(ref) if I == J else got - ref
            worst = max(worst, np.abs(diff).max() / scale)
    assert worst <= tol, worst


@pytest.mark.parametrize("N,B", [(16384, 512), (16384, 1024)])
def test_config2_matches_the_oracle_entry_by_entry(cham, orc, N, B):
    """BASELINE config 2 (N=16384, tile=512; and tile=1024) against the CPU oracle's wave DAG on the same input,
    every entry of the factor: max|dL| / max|L| <= 1e-12 (v6_test.c:56 on the same matrix)."""
    ch = cham
    d = full_desc(ch, N, B)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    _compare_leading_tiles(ch, orc, d, N, B, N // B, 1e-12)
    ch.CHAMELEON_Desc_Destroy(d)


def test_config3_matches_the_oracle_entry_by_entry(cham, orc):
    """BASELINE config 3 (N=32768, tile=512) at full size: EVERY entry of the factor against the CPU oracle's wave DAG
    on the same input (round 4; ~15 s of oracle on the GPU box's 16 cores, 8 GiB), max|dL| / max|L| <= 1e-12."""
    ch = cham
    N, B = 32768, 512
    d = full_desc(ch, N, B)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    _compare_leading_tiles(ch, orc, d, N, B, N // B, 1e-12)
    ch.CHAMELEON_Desc_Destroy(d)


def _trailing_tiles_match_the_oracle(ch, orc, d, N, B, dtype, trail, tol):
    """The END of the run (the chain-bound, counter-linked waves): the trailing `trail` x `trail` tiles of the factor are
    the Cholesky factor of the Schur complement  S = A22 - L21 L21^T.  L21 (the factor's last tile rows, downloaded) and
    A22 (the oracle's generator at that position of the order-N matrix) give S on the host in fp64; the oracle factors
    it; entry by entry against the GPU's trailing tiles."""
    nt, n2 = N // B, trail * B
    first = nt - trail
    S = orc.tile_to_lapack(orc.plgsy_tiles_lower_at(trail, B, float(N), 42, N, first), n2, B)  # A22, lower tiles filled
    S = np.asfortranarray(np.tril(S) + np.tril(S, -1).T)
    if dtype == "f32":
        S = S.astype(np.float32).astype(np.float64)  # (what the fp32 generator stored)
    import torch  # (the checker's fp64 product: test infrastructure; 7.7 TFLOP at config 4 would take minutes on the host)

    Sd = torch.from_numpy(S).cuda()
    L21 = np.empty((n2, B), dtype=np.float64)
    for J in range(first):  # one tile column of L21 at a time: S -= L21(:, J) L21(:, J)^T
        for I in range(trail):
            L21[I * B:(I + 1) * B, :] = d.download_tile(first + I, J)
        Ld = torch.from_numpy(L21).cuda()
        Sd -= Ld @ Ld.T
    S = np.asfortranarray(Sd.cpu().numpy())
    del Sd, Ld
    Lref, info = orc.cholesky_lower(S, B)
    assert info == 0
    scale = np.abs(Lref).max()
    worst = 0.0
    for J in range(trail):
        for I in range(J, trail):
            got = d.download_tile(first + I, first + J).astype(np.float64)
            ref = Lref[I * B:(I + 1) * B, J * B:(J + 1) * B]
            diff = np.tril(got) - ref if I == J else got - ref
            worst = max(worst, np.abs(diff).max() / scale)
    assert worst <= tol, worst


def _leading_block_matches_the_oracle(ch, orc, d, N, B, dtype, lead, tol):
    """The leading `lead` x `lead` block of the factor, entry by entry, against the oracle's factorisation of the leading
    block of the same (order-N) matrix."""
    if dtype == "f32":
        T = orc.plgsy_tiles_lower(lead // B, B, float(N), 42, order=N).astype(np.float32)
        assert orc.tiled_potrf(T, lead // B, B) == 0
        T = T.reshape((lead // B) ** 2, B, B).astype(np.float64)
        nb = lead // B
        scale = max(np.abs(T[I + J * nb]).max() for J in range(nb) for I in range(J, nb))
        for J in range(nb):
            for I in range(J, nb):
                ref, got = T[I + J * nb].T, d.download_tile(I, J).astype(np.float64)
                diff = np.tril(got) - np.tril(ref) if I == J else got - ref
                assert np.abs(diff).max() / scale <= tol, (I, J)
    else:
        _compare_leading_tiles(ch, orc, d, N, B, lead // B, tol)


@pytest.mark.parametrize("N,B,dtype,tol,lead,trail,tol_l", [(65536, 1024, "f64", 1e-13, 8192, 8, 1e-12), (65536, 1024, "f32", 5e-5, 0, 0, 0.0),
                                                            (131072, 1024, "f32", 5e-5, 4096, 4, 1e-4)])
def test_baseline_configs_4_and_5_at_full_size(cham, orc, N, B, dtype, tol, lead, trail, tol_l):
    """BASELINE configs 4 (N=65536 fp64) and 5 (N=131072 fp32, 64 GiB) in full on one GPU, ONE factorisation each:
    residual of the regenerated matrix, positive finite diagonal, untouched strictly-upper tile; the leading block
    entry by entry against the oracle's factorisation of the leading block of the same order-N matrix; the trailing
    tiles -- the chain-bound end of the run -- through the Schur complement (the rest of the factor: the residual)."""
    ch = cham
    dt = ch.ChamRealDouble if dtype == "f64" else ch.ChamRealFloat
    d = full_desc(ch, N, B, dt)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    upper_before = d.download_tile(3, 40)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    assert ch.residual_plgsy(d, float(N), 42) <= tol
    assert np.array_equal(d.download_tile(3, 40), upper_before)
    t = d.download_tile(N // B - 1, N // B - 1)
    assert np.isfinite(np.tril(t)).all() and np.diag(t).min() > 0
    if lead:
        _leading_block_matches_the_oracle(ch, orc, d, N, B, dtype, lead, tol_l)
        _trailing_tiles_match_the_oracle(ch, orc, d, N, B, dtype, trail, tol_l)
    ch.CHAMELEON_Desc_Destroy(d)


@pytest.mark.parametrize("N,B", [(1024, 256), (2048, 512), (1000, 192)])
def test_full_potrf_upper(cham, orc, N, B):
    """ChamUpper (v3 driver's --uplo): A = U^T U in the upper triangle, the strict
    lower triangle of the storage untouched."""
    ch = cham
    d = full_desc(ch, N, B)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamUpperLower, d, 42)
    A = d.to_lapack()
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamUpper, d) == 0
    R = d.to_lapack()
    U = np.triu(R)
    assert np.array_equal(np.tril(R, -1), np.tril(A, -1))
    Lref = np.linalg.cholesky(A)
    assert np.abs(U - Lref.T).max() / np.abs(Lref).max() <= 1e-12
    assert np.linalg.norm(U.T @ U - A) / np.linalg.norm(A) <= 1e-13
    # non-SPD: same info as Lower
    M = A.copy(order="F")
    M[N - 5, N - 5] = -2.0
    d.from_lapack(M)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamUpper, d) == N - 4


def test_residual_inf_norm_matches_numpy(cham):
    """chol_residual_plgsy_inf = ||A - L L^T||_inf / ||A||_inf (v6_test.c:72-86, done right)."""
    ch = cham
    for N, B in ((1024, 256), (1000, 192)):
        d = full_desc(ch, N, B)
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamUpperLower, d, 11)
        A = d.to_lapack()
        assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
        L = np.tril(d.to_lapack())
        t = d.download_tile(1, 0)
        t[3, 4] += 1e-6  # make the residual large enough to compare meaningfully
        d.upload_tile(1, 0, t)
        L[B + 3, 4] += 1e-6
        R = L @ L.T - A
        ref_inf = np.abs(R).sum(axis=1).max() / np.abs(A).sum(axis=1).max()
        ref_fro = np.linalg.norm(R) / np.linalg.norm(A)
        assert abs(ch.residual_plgsy_inf(d, float(N), 11) - ref_inf) <= 1e-3 * ref_inf
        assert abs(ch.residual_plgsy(d, float(N), 11) - ref_fro) <= 1e-3 * ref_fro


def _vm_golden():
    import json

    with open(os.path.join(GOLD, "reference_vm_rel_error.json")) as f:
        return json.load(f)["values"]


def test_validation_block_ops_match_numpy(cham, orc):
    """dlacpy / dlange / dlauum / dgeadd (v6_test.c:51, 74-85) on the device against numpy, on
    a tile-multiple and on a ragged, odd-tile shape; fp64 and fp32."""
    ch = cham
    rng = np.random.default_rng(3)
    for N, B, dt, tol in ((1024, 256, ch.ChamRealDouble, 1e-13), (1000, 192, ch.ChamRealDouble, 1e-13),
                          (520, 128, ch.ChamRealFloat, 2e-5)):
        npdt = np.float64 if dt == ch.ChamRealDouble else np.float32
        A = np.asfortranarray(rng.standard_normal((N, N)).astype(npdt))
        Bm = np.asfortranarray(rng.standard_normal((N, N)).astype(npdt))
        da, db = full_desc(ch, N, B, dt), full_desc(ch, N, B, dt)
        da.from_lapack(A)
        for uplo, f in ((ch.ChamUpperLower, lambda X, Y: X.copy()), (ch.ChamLower, lambda X, Y: np.tril(X) + np.triu(Y, 1)),
                        (ch.ChamUpper, lambda X, Y: np.triu(X) + np.tril(Y, -1))):
            db.from_lapack(Bm)
            assert ch.CHAMELEON_dlacpy_Tile(uplo, da, db) == 0
            assert np.array_equal(db.to_lapack(), f(A, Bm))
        A64 = A.astype(np.float64)
        for norm, ref in ((ch.ChamMaxNorm, np.abs(A64).max()), (ch.ChamOneNorm, np.abs(A64).sum(0).max()),
                          (ch.ChamInfNorm, np.abs(A64).sum(1).max()), (ch.ChamFrobeniusNorm, np.linalg.norm(A64))):
            assert abs(ch.CHAMELEON_dlange_Tile(norm, da) - ref) <= 1e-12 * ref
        db.from_lapack(Bm)
        assert ch.CHAMELEON_dgeadd_Tile(ch.ChamNoTrans, -1.0, da, 0.5, db) == 0
        assert np.abs(db.to_lapack() - (npdt(-1.0) * A + npdt(0.5) * Bm)).max() <= 4 * np.finfo(npdt).eps * 4
        assert ch.CHAMELEON_dlauum_Tile(ch.ChamLower, da) == 0
        got = da.to_lapack()
        want = orc.cham_lauum_lower(A64)
        assert np.array_equal(np.triu(got, 1), np.triu(A, 1))  # strict upper untouched
        assert np.abs(np.tril(got) - np.tril(want)).max() <= tol * N * np.abs(want).max()
    with pytest.raises(ch.CholmiError):
        ch.CHAMELEON_dlacpy_Tile(ch.ChamLower, full_desc(ch, 512, 128), full_desc(ch, 512, 256))
    with pytest.raises(ch.CholmiError):
        ch.CHAMELEON_dlauum_Tile(ch.ChamUpper, full_desc(ch, 512, 128))


@pytest.mark.parametrize("N", [1000, 5000, 8000, 12000, 16000])
def test_v6_validation_as_written_reproduces_the_reference_csv(cham, N):
    """The reference's only recorded numerical outputs: rel_error (column 12) of its bench.csv,
    written by v6_test.c:44-86 for seed 42, bump = N, NB = 128..512 (benchmark.c:76-131).  The
    same call sequence through this library reproduces all of them to the printed digits."""
    from dense_linear_app_amd import driver

    ch = cham
    cases = [(c["NB"], c["rel_error"]) for c in _vm_golden() if c["N"] == N]
    assert len(cases) == 7
    for NB, want in cases:
        dA = full_desc(ch, N, NB)
        dO = full_desc(ch, N, NB)
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, dA, 42)
        ch.CHAMELEON_dlacpy_Tile(ch.ChamUpperLower, dA, dO)
        assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, dA) == 0
        rel = driver.v6_validation_as_written(dA, dO)
        assert "%.2e" % rel == "%.2e" % float(want), (N, NB, rel, want)
        assert ch.residual_plgsy(dA, float(N), 42) <= 1e-13  # the factor itself is at rounding level
        ch.CHAMELEON_Desc_Destroy(dA)
        ch.CHAMELEON_Desc_Destroy(dO)


@pytest.mark.parametrize("N,B,nrhs,dt", [(1024, 256, 256, "f64"), (1536, 512, 1024, "f64"), (1000, 192, 300, "f64"),
                                          (2048, 128, 128, "f64"), (1024, 256, 512, "f32")])
def test_potrs_and_posv_against_numpy(cham, orc, N, B, nrhs, dt):
    """The step after the factor: CHAMELEON_dpotrs_Tile / dposv_Tile, lower, on
    tile-multiple, ragged and odd-tile shapes; checked against numpy.linalg.solve."""
    ch = cham
    cdt = ch.ChamRealDouble if dt == "f64" else ch.ChamRealFloat
    npdt = np.float64 if dt == "f64" else np.float32
    tol = 1e-12 if dt == "f64" else 2e-4
    A = orc.plgsy_matrix(N, float(N), 42)
    rng = np.random.default_rng(11)
    Bm = np.asfortranarray(rng.standard_normal((N, nrhs)))
    X = np.linalg.solve(A, Bm)
    dA = ch.CHAMELEON_Desc_Create(None, cdt, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    dB = ch.CHAMELEON_Desc_Create(None, cdt, B, B, B * B, N, nrhs, 0, 0, N, nrhs, 1, 1)
    # potrf then potrs
    dA.from_lapack(A.astype(npdt))
    dB.from_lapack(Bm.astype(npdt))
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, dA) == 0
    assert ch.CHAMELEON_dpotrs_Tile(ch.ChamLower, dA, dB) == 0
    got = dB.to_lapack().astype(np.float64)
    assert got.shape == (N, nrhs)
    assert np.abs(got - X).max() / np.abs(X).max() <= tol
    assert np.linalg.norm(A @ got - Bm) / (np.linalg.norm(A) * np.linalg.norm(got)) <= (1e-15 if dt == "f64" else 1e-6)
    # posv in one call; a second solve with the same factor gives the same answer
    dA.from_lapack(A.astype(npdt))
    dB.from_lapack(Bm.astype(npdt))
    assert ch.CHAMELEON_dposv_Tile(ch.ChamLower, dA, dB) == 0
    assert np.array_equal(dB.to_lapack().astype(np.float64), got)
    # not positive definite: info > 0, B untouched
    M = A.copy(order="F")
    M[N // 2, N // 2] = -1.0
    dA.from_lapack(M.astype(npdt))
    dB.from_lapack(Bm.astype(npdt))
    assert ch.CHAMELEON_dposv_Tile(ch.ChamLower, dA, dB) == N // 2 + 1
    assert np.array_equal(dB.to_lapack(), Bm.astype(npdt))
    # ChamUpper: factor and solve with U (A = U^T U) on the same storage; same solution to rounding
    if N % B == 0:
        dA.from_lapack(A.astype(npdt))
        dB.from_lapack(Bm.astype(npdt))
        assert ch.CHAMELEON_dpotrf_Tile(ch.ChamUpper, dA) == 0
        assert ch.CHAMELEON_dpotrs_Tile(ch.ChamUpper, dA, dB) == 0
        gotu = dB.to_lapack().astype(np.float64)
        assert np.abs(gotu - X).max() / np.abs(X).max() <= tol
        U = np.triu(dA.to_lapack().astype(np.float64))
        assert np.linalg.norm(U.T @ U - A) / np.linalg.norm(A) <= (1e-13 if dt == "f64" else 5e-5)


@pytest.mark.parametrize("env", [{"CHOLMI_PAIR_FACTOR": "0"}, {"CHOLMI_PAIR_FACTOR": "1000"},
                                 # pairs entered behind a wave launched as near / far halves (wave 0 stays plain), left at the end
                                 {"CHOLMI_PAIR_FACTOR": "0", "CHOLMI_HALVES_MAX_ROUNDS": "1000", "CHOLMI_PIPE_FACTOR": "0"},
                                 # the chain-bound form (device-side counters): off (events only), from the
                                 # first wave on, and entered late (event-linked waves first, then counters)
                                 {"CHOLMI_DEVICE_FLAGS": "0"}, {"CHOLMI_PIPE_FACTOR": "100", "CHOLMI_PAIR_FACTOR": "1000"},
                                 {"CHOLMI_PIPE_FACTOR": "0.02"},
                                 # near / far halves of the update on two streams, boundary moving every few waves
                                 {"CHOLMI_PIPE_FACTOR": "0", "CHOLMI_HALVES_MAX_ROUNDS": "1000", "CHOLMI_PAIR_FACTOR": "1000"},
                                 {"CHOLMI_HALVES_MAX_ROUNDS": "0", "CHOLMI_PIPE_FACTOR": "0"},
                                 # counters: every grid behind a gate kernel / every grid polling itself
                                 {"CHOLMI_POLL_MAX_WGS": "0"}, {"CHOLMI_POLL_MAX_WGS": "100000"},
                                 # counter-linked waves from the first one on: without the near column (column k+2 as a launch
                                 # of its own) and the latency form of column k+1; with both for every wave
                                 {"CHOLMI_PIPE_FACTOR": "100", "CHOLMI_PAIR_FACTOR": "1000", "CHOLMI_NEAR_FACTOR": "0", "CHOLMI_U1_SMALL": "0"},
                                 {"CHOLMI_PIPE_FACTOR": "100", "CHOLMI_PAIR_FACTOR": "1000", "CHOLMI_NEAR_FACTOR": "100", "CHOLMI_U1_SMALL": "64"},
                                 # ... entered behind waves launched as wider near / far halves
                                 {"CHOLMI_PIPE_FACTOR": "0.3", "CHOLMI_NEAR_FACTOR": "100", "CHOLMI_HALVES_MAX_ROUNDS": "1000", "CHOLMI_PAIR_FACTOR": "1000"}])
def test_walker_schedule_variants_match_the_oracle(env, orc):
    """The walker picks its schedule by size (two panels per pass only while a wave's update is long, and so
    on), so at oracle-sized problems the default run never enters some of them.  Each variant forced through
    its environment switch in a fresh process (the switches are read once): factor vs the oracle's, element
    by element, fp64 and fp32."""
    import subprocess
    import sys
    import tempfile

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    N, B = 3072, 256  # 12 tiles per side
    with tempfile.TemporaryDirectory() as tmp:
        code = (
            "import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from dense_linear_app_amd import chameleon as ch\n"
            "ch.CHAMELEON_Init(1, 1)\n"
            "for name, dt in (('d', ch.ChamRealDouble), ('s', ch.ChamRealFloat)):\n"
            "    d = ch.CHAMELEON_Desc_Create(None, dt, %d, %d, %d, %d, %d, 0, 0, %d, %d, 1, 1)\n"
            "    ch.CHAMELEON_dplgsy_Tile(float(%d), ch.ChamLower, d, 42)\n"
            "    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)\n"
            "    np.save(%r + '/L' + name + '.npy', d.to_lapack()); print(name, info)\n"
        ) % (root, B, B, B * B, N, N, N, N, N, tmp)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0 and "d 0" in r.stdout and "s 0" in r.stdout, (r.stdout, r.stderr[-2000:])
        Ld, Ls = np.load(tmp + "/Ld.npy"), np.load(tmp + "/Ls.npy")
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = np.tril(orc.tile_to_lapack(T, N, B))
    assert np.abs(np.tril(Ld) - Lref).max() / np.abs(Lref).max() <= 1e-12
    assert np.abs(np.tril(Ls).astype(np.float64) - Lref).max() / np.abs(Lref).max() <= 1e-4


FLOW_ALL = {"CHOLMI_FLOW_FACTOR": "100", "CHOLMI_PIPE_FACTOR": "100", "CHOLMI_PAIR_FACTOR": "1000"}


@pytest.mark.parametrize("N,B,env", [
    (3072, 512, FLOW_ALL),                                         # four 128-blocks per tile, every wave in flow form
    (3072, 384, FLOW_ALL),                                         # three
    (3072, 512, dict(FLOW_ALL, CHOLMI_FLOW_FENCES="1")),           # ... with release / acquire fences around every hand-off
    (4096, 1024, dict(FLOW_ALL, CHOLMI_FLOW_NBM="2:8")),           # eight blocks per tile (off by default: measured slower)
    (2048, 256, dict(FLOW_ALL, CHOLMI_FLOW_NBM="2:4")),            # two
    (4096, 512, {"CHOLMI_FLOW_FACTOR": "0.05"}),                   # entered late: event-linked, counter-linked, then flow waves
    (4096, 512, {}),                                               # the default rule: chain-bound from wave 0 on -> flow
    # flow waves behind PAIRED waves: no counter to poll, the row-slab kernel is ordered by its join event alone (advisor, round 4)
    (3584, 512, {"CHOLMI_PAIR_FACTOR": "0", "CHOLMI_FLOW_FACTOR": "0.1"}),
])
def test_flow_form_of_the_tile_potrf_matches_the_oracle(N, B, env, orc):
    """k_flow_factor / k_flow_rows (round 4): the tile POTRF of a counter-linked wave as two persistent launches whose
    workgroups hand panels of 16 columns to each other through polled counters and write-through stores.  Forced on for
    every wave (and entered late, and by its default rule) in a fresh process; factor vs the oracle's, element by
    element, fp64 and fp32; the library must say that the form was actually used."""
    import subprocess
    import sys
    import tempfile

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    with tempfile.TemporaryDirectory() as tmp:
        code = (
            "import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from dense_linear_app_amd import chameleon as ch\n"
            "from dense_linear_app_amd._lib import lib\n"
            "ch.CHAMELEON_Init(1, 1)\n"
            "for name, dt in (('d', ch.ChamRealDouble), ('s', ch.ChamRealFloat)):\n"
            "    d = ch.CHAMELEON_Desc_Create(None, dt, %d, %d, %d, %d, %d, 0, 0, %d, %d, 1, 1)\n"
            "    ch.CHAMELEON_dplgsy_Tile(float(%d), ch.ChamLower, d, 42)\n"
            "    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)\n"
            "    np.save(%r + '/L' + name + '.npy', d.to_lapack()); print(name, info, 'flow_waves', lib().chol_debug_flow_waves())\n"
        ) % (root, B, B, B * B, N, N, N, N, N, tmp)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0 and "d 0" in r.stdout and "s 0" in r.stdout, (r.stdout, r.stderr[-2000:])
        used = [int(ln.split()[-1]) for ln in r.stdout.splitlines() if "flow_waves" in ln]
        assert len(used) == 2 and min(used) >= 1, r.stdout
        Ld, Ls = np.load(tmp + "/Ld.npy"), np.load(tmp + "/Ls.npy")
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = np.tril(orc.tile_to_lapack(T, N, B))
    assert np.abs(np.tril(Ld) - Lref).max() / np.abs(Lref).max() <= 1e-12
    assert np.abs(np.tril(Ls).astype(np.float64) - Lref).max() / np.abs(Lref).max() <= 1e-4


def test_flow_form_reports_a_failed_pivot_and_ends(orc):
    """A pivot that is not positive (and a NaN) inside a flow-form wave: the diagonal-block workgroup raises the flow's
    abort word, every polling workgroup of both launches leaves, the counters the other streams poll still come, and
    chol_potrf_tile returns LAPACK's info -- at once, not after a poll's bound."""
    import subprocess
    import sys
    import time

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from dense_linear_app_amd import chameleon as ch\n"
        "from dense_linear_app_amd._lib import lib\n"
        "ch.CHAMELEON_Init(1, 1)\n"
        "N, B = 3072, 512\n"
        "for what, g, v in (('neg', 1300, -5.0), ('nan', 1801, float('nan')), ('ok', 0, None)):\n"
        "    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)\n"
        "    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)\n"
        "    if v is not None:\n"
        "        t = d.download_tile(g // B, g // B); t[g %% B, g %% B] = v; d.upload_tile(g // B, g // B, t)\n"
        "    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)\n"
        "    print(what, info, 'flow_waves', lib().chol_debug_flow_waves())\n"
    ) % root
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **FLOW_ALL), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    out = dict((ln.split()[0], int(ln.split()[1])) for ln in r.stdout.splitlines() if "flow_waves" in ln)
    assert out == {"neg": 1301, "nan": 1802, "ok": 0}, r.stdout
    assert time.time() - t0 < 120


def test_two_processes_factorising_on_one_gpu_do_not_starve_each_other():
    """The counter-linked form of a chain-bound wave launches consumer kernels ahead of time and lets them poll.
    Large polling grids once held so much LDS that the kernel they waited for found no CU -- a deadlock until the
    poll's bound (info = INT_MAX - 1), seen with two processes time-sliced on one GPU at exactly this shape.
    Grids above a few dozen workgroups now wait behind a one-wave gate kernel instead (kernels.hip: k_sem_gate)."""
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from dense_linear_app_amd import chameleon as ch\n"
        "ch.CHAMELEON_Init(1, 1)\n"
        "for N, B, reps in ((8192, 1024, 12), (4096, 512, 30)):\n"
        "    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)\n"
        "    for r in range(reps):\n"
        "        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 7 + r)\n"
        "        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)\n"
        "        res = ch.residual_plgsy(d, float(N), 7 + r)\n"
        "        assert info == 0 and res <= 1e-13, (N, B, r, info, res)\n"
        "print('ok')\n"
    ) % root
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0 and "ok" in out, (out, err[-2000:])
