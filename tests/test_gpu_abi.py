"""The C ABI's argument checking and descriptor semantics on the GPU box: LAPACK-style
negative statuses for bad arguments, CHOL_ERR_NOT_SUPPORTED for valid-but-uncovered
Chameleon usage (never a wrong answer), randomised shapes/scalars against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu


def _desc(L, a, dt=3, **kw):
    B = a.shape[0]
    args = dict(mb=B, nb=B, bsiz=B * B, lm=B, ln=B, i=0, j=0, m=B, n=B, p=1, q=1)
    args.update(kw)
    h = C.c_void_p()
    rc = L.chol_desc_create(C.byref(h), a.ctypes.data, dt, args["mb"], args["nb"], args["bsiz"], args["lm"],
                            args["ln"], args["i"], args["j"], args["m"], args["n"], args["p"], args["q"])
    return rc, h


def test_desc_create_argument_positions(cham):
    from dense_linear_app_amd._lib import lib

    L = lib()
    a = np.zeros((8, 8), order="F")
    assert _desc(L, a)[0] == 0
    assert _desc(L, a, dt=7)[0] == -3
    assert _desc(L, a, mb=0)[0] == -4
    assert _desc(L, a, nb=-1)[0] == -5
    assert _desc(L, a, bsiz=63)[0] == -6
    assert _desc(L, a, lm=0)[0] == -7
    assert _desc(L, a, ln=0)[0] == -8
    assert _desc(L, a, m=9)[0] == -11
    assert _desc(L, a, n=9)[0] == -12
    assert _desc(L, a, p=0)[0] == -13
    assert _desc(L, a, q=0)[0] == -14
    rc, hv = _desc(L, a, i=2, m=6)  # a sub-matrix view that starts inside a tile: served (mirrored) since round 4
    assert rc == 0 and L.chol_desc_destroy(C.byref(hv)) == 0
    assert _desc(L, a, lm=12, ln=12, m=8, n=8)[0] == -104  # ... over a user matrix that is not made of whole tiles: not covered
    assert _desc(L, a, p=2)[0] == -104       # p*q must match the number of ranks
    assert b"chol_set_rank" in L.chol_last_error()


def test_ops_reject_null_and_mismatched_descriptors(cham):
    from dense_linear_app_amd._lib import lib

    ch, L = cham, lib()
    a8, a16 = np.eye(8, order="F"), np.eye(16, order="F")
    _, d8 = _desc(L, a8)
    _, d16 = _desc(L, a16)
    assert L.chol_potrf_tile(ch.ChamLower, None) == -2
    assert L.chol_potrf_tile(999, d8) == -1
    assert L.chol_trsm_tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, None, d8) == -6
    assert L.chol_trsm_tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, d8, None) == -7
    assert L.chol_trsm_tile(0, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, d8, d8) == -1
    assert L.chol_trsm_tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, d8, d16) == -104
    assert L.chol_syrk_tile(ch.ChamLower, ch.ChamNoTrans, -1.0, d8, 1.0, None) == -6
    assert L.chol_syrk_tile(ch.ChamUpper, ch.ChamNoTrans, -1.0, d8, 1.0, d8) == -104
    assert L.chol_gemm_tile(ch.ChamNoTrans, ch.ChamTrans, -1.0, d8, None, 1.0, d8) == -5
    assert L.chol_gemm_tile(ch.ChamNoTrans, ch.ChamTrans, -1.0, d8, d16, 1.0, d8) == -104
    f8 = np.eye(8, dtype=np.float32, order="F")
    _, df = _desc(L, f8, dt=ch.ChamRealFloat)
    assert L.chol_gemm_tile(ch.ChamNoTrans, ch.ChamTrans, -1.0, d8, df, 1.0, d8) == -104  # mixed types
    for h in (d8, d16, df):
        assert L.chol_desc_destroy(C.byref(h)) == 0
    assert L.chol_desc_destroy(C.byref(C.c_void_p())) == -1


def test_whole_matrix_descriptor_restrictions(cham):
    ch = cham
    # ragged order / non-128 tiles are served from a library-owned padded image only
    buf = np.zeros(1024 * 1024)
    with pytest.raises(ch.CholmiError, match="ragged"):
        ch.CHAMELEON_Desc_Create(buf, ch.ChamRealDouble, 256, 256, 256 * 256, 1000, 1000, 0, 0, 1000, 1000, 1, 1)
    with pytest.raises(ch.CholmiError, match="mb == nb"):
        ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, 256, 128, 256 * 128, 1024, 1024, 0, 0, 1024, 1024, 1, 1)
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, 256, 256, 256 * 256, 512, 1024, 0, 0, 512, 1024, 1, 1)
    with pytest.raises(ch.CholmiError, match="not square"):
        ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)


@pytest.mark.parametrize("N,NB", [(1000, 192), (1000, 128), (5000, 448), (2000, 320), (777, 256)])
def test_reference_sweep_shapes_ragged_and_odd_tiles(cham, N, NB):
    """The VM sweep's grid (benchmark.c:76-80: N in {1000, 5000, ...}, NB = 128..512 step 64):
    order not a multiple of the tile, tile not a multiple of 128.  Edge tiles are smaller, not
    zero-padded (v6_test.c:44 descriptor semantics); the library keeps a padded image inside."""
    ch = cham
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, NB, NB, NB * NB, N, N, 0, 0, N, N, 1, 1)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamUpperLower, d, 42)
    A = d.to_lapack()
    assert A.shape == (N, N) and np.array_equal(A, A.T) and np.diag(A).min() > N - 1
    from oracle import oracle as orc

    assert A[N - 1, 3] == orc.plgsy_entry(float(N), 42, N, N - 1, 3)  # same generator, global indices
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    L = np.tril(d.to_lapack())
    Lref = np.linalg.cholesky(A)
    assert np.abs(L - Lref).max() / np.abs(Lref).max() <= 1e-12
    assert ch.residual_plgsy(d, float(N), 42) <= 1e-13
    # a user matrix through Lapack_to_Tile, and the pivot index of a non-SPD one in GLOBAL numbering
    M = A.copy(order="F")
    bad = N - 7
    M[bad, bad] = -1.0
    d.from_lapack(M)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == bad + 1
    t = d.download_tile(d.mt - 1, 0)
    rows = N - (d.mt - 1) * NB
    assert t.shape == (NB, NB) and not t[rows:, :].any()


def test_host_resident_tiled_matrix_is_staged(cham, orc):
    """A multi-tile descriptor over a HOST buffer in tile layout: staged through HBM as a whole."""
    ch = cham
    N, B = 1024, 256
    A = orc.reference_input(N)
    T = orc.lapack_to_tile(A, B)
    d = ch.CHAMELEON_Desc_Create(T, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    L = np.tril(orc.tile_to_lapack(T, N, B))
    Lref, _ = orc.cholesky_lower(A, B)
    assert np.abs(L - Lref).max() / np.abs(Lref).max() <= 1e-12


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(B=st.integers(1, 300), alpha=st.floats(-2, 2), beta=st.floats(-2, 2), seed=st.integers(0, 2 ** 31))
def test_gemm_syrk_random_shapes_and_scalars(cham, orc, B, alpha, beta, seed):
    ch = cham
    rng = np.random.default_rng(seed)
    A = np.asfortranarray(rng.uniform(-1, 1, (B, B)))
    Bm = np.asfortranarray(rng.uniform(-1, 1, (B, B)))
    C0 = np.asfortranarray(rng.uniform(-1, 1, (B, B)))
    mk = lambda a: ch.CHAMELEON_Desc_Create(a, ch.ChamRealDouble, B, B, B * B, B, B, 0, 0, B, B, 1, 1)
    tol = 16 * max(B, 8) * np.finfo(float).eps * (abs(alpha) * B + abs(beta) + 1)
    Cg = C0.copy(order="F")
    assert ch.CHAMELEON_dgemm_Tile(ch.ChamNoTrans, ch.ChamTrans, alpha, mk(A), mk(Bm), beta, mk(Cg)) == 0
    assert np.abs(Cg - orc.dgemm(A, Bm, C0, alpha, beta)).max() <= tol
    Cs = C0.copy(order="F")
    assert ch.CHAMELEON_dsyrk_Tile(ch.ChamLower, ch.ChamNoTrans, alpha, mk(A), beta, mk(Cs)) == 0
    assert np.abs(np.tril(Cs) - np.tril(orc.dsyrk(A, C0, alpha, beta))).max() <= tol
    assert np.array_equal(np.triu(Cs, 1), np.triu(C0, 1))


@settings(max_examples=15, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(B=st.integers(1, 260), seed=st.integers(0, 2 ** 31))
def test_potrf_trsm_random_spd(cham, orc, B, seed):
    ch = cham
    rng = np.random.default_rng(seed)
    M = rng.uniform(-1, 1, (B, B))
    S = np.asfortranarray(M @ M.T + B * np.eye(B))
    mk = lambda a: ch.CHAMELEON_Desc_Create(a, ch.ChamRealDouble, B, B, B * B, B, B, 0, 0, B, B, 1, 1)
    L = S.copy(order="F")
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, mk(L)) == 0
    Lref = np.linalg.cholesky(S)
    assert np.abs(np.tril(L) - Lref).max() <= 64 * max(B, 8) * np.finfo(float).eps * np.abs(Lref).max()
    X = np.asfortranarray(rng.uniform(-1, 1, (B, B)))
    X0 = X.copy()
    assert ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, mk(L), mk(X)) == 0
    assert np.abs(X @ Lref.T - X0).max() <= 256 * max(B, 8) * np.finfo(float).eps * max(1.0, np.abs(X).max()) * np.abs(Lref).max()


def test_plain_c_driver_reproduces_reference_outputs():
    """examples/v6_driver (C99, links libcholmi.so only): same arguments and output lines as the
    reference's v6_test.c; its validation line equals the value the reference recorded."""
    import json
    import re
    import subprocess

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = os.path.join(root, "examples", "v6_driver")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "all"])
    with open(os.path.join(root, "tests", "golden", "reference_vm_rel_error.json")) as f:
        gold = {(c["N"], c["NB"]): c["rel_error"] for c in json.load(f)["values"]}
    for N, NB in ((1000, 128), (5000, 320), (8000, 512)):
        args = [1, 1, N, NB, NB, NB, NB * NB, N, N, 0, 0, N, N, 1, 1, 42]
        r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert re.search(r"Performance: [0-9.]+ Gflop/s", r.stdout)
        m = re.search(r"\|\|A - LL\^T\|\|_inf / \|\|A\|\|_inf = ([0-9.e+-]+)", r.stdout)
        assert m and m.group(1) == "%.2e" % float(gold[N, NB]), r.stdout
        m = re.search(r"\|\|A - L L\^T\|\|_F / \|\|A\|\|_F = ([0-9.e+-]+)", r.stdout)
        assert m and float(m.group(1)) <= 1e-13, r.stdout


def test_tile_above_workspace_capacity_is_refused(cham):
    """The context holds the inverses of 32 diagonal 128-blocks (tiles up to 4096 in fp64): a larger
    tile must be refused by every entry that would write that workspace -- never an out-of-bounds
    device write (B comes straight from the worker's JSON payload)."""
    from dense_linear_app_amd._lib import lib

    ch, L = cham, lib()
    B = 4224
    a = np.zeros((B, B), order="F")
    a[np.arange(B), np.arange(B)] = 2.0
    rc, d = _desc(L, a)
    assert rc == 0
    before = a.copy()
    assert L.chol_potrf_tile(ch.ChamLower, d) == -104 and b"4096" in L.chol_last_error()
    assert L.chol_trsm_tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, d, d) == -104
    assert np.array_equal(a, before)
    assert L.chol_desc_destroy(C.byref(d)) == 0
    # the largest tile that fits still works
    B = 4096
    a = np.zeros((B, B), order="F")
    a[np.arange(B), np.arange(B)] = 4.0
    rc, d = _desc(L, a)
    assert rc == 0 and L.chol_potrf_tile(ch.ChamLower, d) == 0
    assert np.array_equal(np.diag(a), np.full(B, 2.0))
    assert L.chol_desc_destroy(C.byref(d)) == 0


@pytest.mark.parametrize("n,mb", [(150, 200), (100, 128), (300, 512)])
def test_matrix_smaller_than_its_single_tile(cham, orc, n, mb):
    """mt = nt = 1 with lm < mb (a ragged edge tile that is also the only tile): served from the
    padded library-owned image like every other ragged shape, refused over a user buffer."""
    ch = cham
    rng = np.random.default_rng(n)
    M = rng.uniform(-0.5, 0.5, (n, n))
    A = np.asfortranarray(M @ M.T + n * np.eye(n))
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, mb, mb, mb * mb, n, n, 0, 0, n, n, 1, 1)
    d.from_lapack(A)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    out = d.to_lapack()
    Lref, info = orc.dpotrf(A)
    assert info == 0
    assert np.abs(np.tril(out) - np.tril(Lref)).max() <= 16 * n * np.finfo(float).eps * np.abs(Lref).max()
    assert np.array_equal(np.triu(out, 1), np.triu(A, 1))
    bad = A.copy(order="F")
    bad[n - 3, n - 3] = -1.0
    d.from_lapack(bad)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == n - 2
    with pytest.raises(ch.CholmiError, match="ragged"):
        ch.CHAMELEON_Desc_Create(np.zeros(mb * mb), ch.ChamRealDouble, mb, mb, mb * mb, n, n, 0, 0, n, n, 1, 1)


def test_submatrix_views_on_library_owned_descriptors(cham, orc):
    """CHAMELEON_Desc_Create(..., lm, ln, i, j, m, n, ...) with (i, j, m, n) != (0, 0, lm, ln): the
    reference passes ioff, joff, m, n from argv (v6_test.c:24-25, 44-45).  Library-owned storage and
    tile-aligned offsets: generator, factorisation and layout conversion work in view coordinates."""
    ch = cham
    mb, lm, i0, m = 256, 2048, 512, 1024
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, mb, mb, mb * mb, lm, lm, i0, i0, m, m, 1, 1)
    assert (d.mt, d.nt) == (4, 4)
    ch.CHAMELEON_dplgsy_Tile(float(m), ch.ChamUpperLower, d, 42)
    A = d.to_lapack()
    assert A.shape == (m, m)
    # Chameleon's dplgsy on a view: entry (r, c) of the view of an order-m matrix
    assert np.array_equal(A, orc.plgsy_matrix(m, float(m), 42))
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    Lref, info = orc.cholesky_lower(A, mb)
    assert info == 0
    assert np.abs(np.tril(d.to_lapack()) - Lref).max() <= 1e-12 * np.abs(Lref).max()
    assert ch.residual_plgsy(d, float(m), 42) <= 1e-13
    # a ragged view that runs to the end of the matrix, rectangular parent
    d2 = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, 192, 192, 192 * 192, 1000, 1200, 384, 576, 616, 616, 1, 1)
    ch.CHAMELEON_dplgsy_Tile(616.0, ch.ChamLower, d2, 7)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d2) == 0
    assert ch.residual_plgsy(d2, 616.0, 7) <= 1e-13
    # an offset inside a tile (round 4): over library-owned storage nothing outside the view is observable, so the view is
    # the same m x m matrix of its own wherever it starts
    d3 = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, mb, mb, mb * mb, lm, lm, 100, 37, m, m, 1, 1)
    ch.CHAMELEON_dplgsy_Tile(float(m), ch.ChamUpperLower, d3, 42)
    assert np.array_equal(d3.to_lapack(), A)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d3) == 0
    assert np.abs(np.tril(d3.to_lapack()) - Lref).max() <= 1e-12 * np.abs(Lref).max()
    # over a user buffer the USER's matrix must be made of whole tiles (test_sub_matrix_view_* cover the served cases)
    with pytest.raises(ch.CholmiError, match="user buffer"):
        ch.CHAMELEON_Desc_Create(np.zeros(2000 * 2000), ch.ChamRealDouble, mb, mb, mb * mb, 2000, 2000, i0, i0, m, m, 1, 1)


def test_v3_long_option_driver(cham, orc):
    """The reference's long-option front end (v3_script_cholesky_x_arg_gpt.c): --dtyp d|s, --uplo L|U,
    --bump, offsets; every option required; same output lines; Python and plain-C forms."""
    import io
    import re
    import subprocess

    from dense_linear_app_amd import driver

    base = {"N": 2048, "NB": 256, "ncpu": 1, "ngpu": 1, "mat": "none", "dtyp": "d", "mb": 256, "nb": 256,
            "bsiz": 65536, "lm": 2048, "ln": 2048, "i": 0, "j": 0, "m": 2048, "n": 2048, "p": 1, "q": 1,
            "bump": 2048, "uplo": "L", "seed": 51}

    def args(**kw):
        d = dict(base, **kw)
        out = []
        for k, v in d.items():
            out += [f"--{k}", str(v)]
        return out

    for kw in ({}, {"uplo": "U"}, {"dtyp": "s"}, {"i": 512, "j": 512, "m": 1024, "n": 1024, "bump": 1024},
               {"mat": "user"}, {"mat": "user", "i": 512, "j": 256, "m": 1024, "n": 1024, "bump": 1024}):
        out, err = io.StringIO(), io.StringIO()
        assert driver.v3_test(args(**kw), out=out, err=err) == 0, err.getvalue()
        text = out.getvalue()
        assert re.search(r"^N=2048 NB=256 ncpu=1 ngpu=1 p=1 q=1 bump=\d+ uplo=12[12] seed=51$", text, re.M), text
        assert re.search(r"^Time: \d+\.\d{6} s$", text, re.M) and re.search(r"^Performance: [0-9.]+ Gflop/s$", text, re.M)
    out, err = io.StringIO(), io.StringIO()
    assert driver.v3_test(args()[:-2], out=out, err=err) == 1 and "all options are required" in err.getvalue()
    assert driver.v3_test(args(uplo="X"), out=out, err=err) == 1
    assert driver.v3_test(args(dtyp="z"), out=out, err=err) == 1
    assert driver.v3_test(args(i=512, m=2048), out=out, err=err) == 1  # sub-matrix outside lm
    err = io.StringIO()
    assert driver.v3_test(args(bump=0), out=io.StringIO(), err=err) == 1  # not SPD: info != 0
    assert "bump==0" in err.getvalue() and "Erreur dans CHAMELEON_dpotrf_Tile" in err.getvalue()
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = os.path.join(root, "examples", "v3_driver")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "examples"), "all"])
    r = subprocess.run([exe] + args(uplo="U", dtyp="s"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert re.search(r"^N=2048 NB=256 ncpu=1 ngpu=1 p=1 q=1 bump=2048 uplo=121 seed=51$", r.stdout, re.M), r.stdout
    r = subprocess.run([exe] + args()[:-2], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "all options are required" in r.stderr


def test_desc_version_tag_reuses_the_block_inverses_of_the_factored_tile(cham, orc):
    """chol_desc_set_version: POTRF on a tagged in-place device tile keeps the tile's block inverses; TRSM with an L of
    the same buffer AND tag uses them, any other tag (or none) recomputes them from L -- same solution to rounding."""
    import torch

    ch = cham
    B = 512
    A = orc.reference_input(B)
    rng = np.random.default_rng(3)
    X0 = np.asfortranarray(rng.standard_normal((B, B)))
    dl = torch.from_numpy(A.ravel(order="F").copy()).cuda()
    dL = ch.CHAMELEON_Desc_Create(dl, ch.ChamRealDouble, B, B, B * B, B, B, 0, 0, B, B, 1, 1)
    dL.set_version(0x1234abcd)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, dL) == 0
    outs = []
    for tag in (0x1234abcd, 0x777, 0):
        dx = torch.from_numpy(X0.ravel(order="F").copy()).cuda()
        dX = ch.CHAMELEON_Desc_Create(dx, ch.ChamRealDouble, B, B, B * B, B, B, 0, 0, B, B, 1, 1)
        dL.set_version(tag)
        assert ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, dL, dX) == 0
        outs.append(dx.cpu().numpy().reshape((B, B), order="F").copy())
        ch.CHAMELEON_Desc_Destroy(dX)
    L = np.tril(dl.cpu().numpy().reshape((B, B), order="F"))
    ref = orc.dtrsm(L, X0)
    for o in outs:
        assert np.abs(o - ref).max() <= 16 * B * 2.0 ** -52 * np.abs(ref).max()
    assert np.array_equal(outs[1], outs[2])  # both recomputed from L
    ch.CHAMELEON_Desc_Destroy(dL)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_tile_batches_through_the_raw_abi(cham, dtype):
    """chol_potrf_batch / chol_tile_batch on lists of device pointers, both precisions, several tasks per call:
    every task equals the numpy result of the same operation, inputs are left untouched (private copies), a tile
    that is not positive definite reports its own info and leaves the others alone."""
    import torch

    from dense_linear_app_amd._lib import lib

    ch, L = cham, lib()
    B, n = 256, 3
    npdt, tdt, cdt = (np.float64, torch.float64, ch.ChamRealDouble) if dtype == "f64" else (np.float32, torch.float32, ch.ChamRealFloat)
    tol = 1e-12 if dtype == "f64" else 2e-4
    rng = np.random.default_rng(5)

    def dev(a):
        return torch.from_numpy(np.asfortranarray(a).ravel(order="F").copy()).cuda()

    def host(t):
        return t.cpu().numpy().reshape((B, B), order="F").astype(np.float64)

    def ptrs(ts):
        return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    spd = []
    for q in range(n):
        M = rng.standard_normal((B, B))
        spd.append((M @ M.T + B * np.eye(B)).astype(npdt))
    spd[1][7, 7] = -1.0  # task 1 is not positive definite
    a_in = [dev(a) for a in spd]
    a_out = [torch.zeros(B * B, dtype=tdt, device="cuda") for _ in range(n)]
    slots = (C.c_int * n)()
    assert L.chol_potrf_batch(cdt, B, n, ptrs(a_in), ptrs(a_out), None, slots, 1) == 0, L.chol_last_error()
    infos = []
    for q in range(n):
        v = C.c_int()
        assert L.chol_batch_info(slots[q], C.byref(v)) == 0
        infos.append(v.value)
    assert infos == [0, 8, 0]
    for q in (0, 2):
        Lq = np.tril(host(a_out[q]))
        A = spd[q].astype(np.float64)
        assert np.linalg.norm(Lq @ Lq.T - A) / np.linalg.norm(A) <= (1e-14 if dtype == "f64" else 1e-5)
        assert np.array_equal(host(a_in[q]), A)  # the input was copied, not factored in place
    # TRSM / SYRK / GEMM: three tasks each
    Lt = [np.tril(host(a_out[0])).astype(npdt)] * n
    Cs = [rng.standard_normal((B, B)).astype(npdt) for _ in range(n)]
    As = [rng.standard_normal((B, B)).astype(npdt) for _ in range(n)]
    Bs = [rng.standard_normal((B, B)).astype(npdt) for _ in range(n)]
    dC, dA, dB, dL = [dev(c) for c in Cs], [dev(a) for a in As], [dev(b) for b in Bs], [dev(Lt[0])] * n
    out = torch.zeros(n * B * B, dtype=tdt, device="cuda")
    outs = [out[q * B * B:(q + 1) * B * B] for q in range(n)]
    for code, a_list, b_list in ((1, dL, None), (2, dA, None), (3, dA, dB)):
        out.zero_()
        rc = L.chol_tile_batch(code, cdt, B, n, ptrs(dC), ptrs(a_list), ptrs(b_list) if b_list else None, ptrs(outs), None, 0)
        assert rc == 0, L.chol_last_error()
        for q in range(n):
            got, Cq, Aq, Bq = host(outs[q]), Cs[q].astype(np.float64), As[q].astype(np.float64), Bs[q].astype(np.float64)
            if code == 1:
                ref = np.linalg.solve(Lt[0].astype(np.float64), Cq.T).T  # X L^T = C
            elif code == 2:
                ref = Cq - Aq @ Aq.T
                ref[np.triu_indices(B, 1)] = Cq[np.triu_indices(B, 1)]  # strict upper triangle: the copy
            else:
                ref = Cq - Aq @ Bq.T
            assert np.abs(got - ref).max() <= tol * B * max(1.0, np.abs(ref).max()), (code, q)
            assert np.array_equal(host(dC[q]), Cq)
    # CHOL_BATCH_UPDATE: SYRK and GEMM tasks in ONE launch (b == NULL marks the SYRK task), out of place
    out.zero_()
    mixed = (C.c_void_p * n)(dB[0].data_ptr(), None, dB[2].data_ptr())
    assert L.chol_tile_batch(4, cdt, B, n, ptrs(dC), ptrs(dA), mixed, ptrs(outs), None, 0) == 0, L.chol_last_error()
    for q in range(n):
        Cq, Aq, Bq = Cs[q].astype(np.float64), As[q].astype(np.float64), Bs[q].astype(np.float64)
        ref = Cq - Aq @ (Aq if q == 1 else Bq).T
        if q == 1:
            ref[np.triu_indices(B, 1)] = Cq[np.triu_indices(B, 1)]
        assert np.abs(host(outs[q]) - ref).max() <= tol * B * max(1.0, np.abs(ref).max()), q
        if q == 1:
            assert np.array_equal(np.triu(host(outs[q]), 1), np.triu(Cq, 1))  # copied bit for bit
    assert L.chol_tile_batch(3, cdt, B, n, ptrs(dC), ptrs(dA), mixed, ptrs(outs), None, 0) < 0  # a GEMM task needs its b
    assert L.chol_tile_batch(9, cdt, B, n, ptrs(dC), ptrs(dA), None, ptrs(outs), None, 0) < 0
    assert L.chol_tile_batch(3, cdt, 100, n, ptrs(dC), ptrs(dA), ptrs(dB), ptrs(outs), None, 0) == -104


@pytest.mark.parametrize("mb", [256, 192])
@pytest.mark.parametrize("where", ["host", "device"])
def test_sub_matrix_view_over_a_user_buffer(cham, orc, where, mb):
    """CHAMELEON_Desc_Create(mat != NULL, ..., i, j, m, n): a tile-aligned view of the user's tile matrix (v3 driver:
    --mat user with offsets).  The library mirrors the view through a device image: the view's tiles are factored in
    the user's buffer, every other tile of it stays bit for bit what it was.  mb = 192 (the reference's NB sweep): the
    image's tiles are padded to 256, the user's are not -- the two sides are addressed with their own tile strides."""
    import torch

    ch = cham
    lt, oi, oj, vt = 5, 1, 2, 3  # user matrix: 5 x 5 tiles; view: 3 x 3 tiles starting at tile (1, 2)
    lm = lt * mb
    m = vt * mb
    rng = np.random.default_rng(9)
    user = rng.standard_normal(lt * lt * mb * mb)  # tile layout, every tile noise
    A = orc.plgsy_matrix(m, float(m), 7)
    buf = user.copy() if where == "host" else torch.from_numpy(user.copy()).cuda()
    d = ch.CHAMELEON_Desc_Create(buf, ch.ChamRealDouble, mb, mb, mb * mb, lm, lm, oi * mb, oj * mb, m, m, 1, 1)
    d.from_lapack(A)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    L = np.tril(d.to_lapack())
    Lref, info = orc.cholesky_lower(A, mb)
    assert info == 0 and np.abs(L - Lref).max() <= 1e-12 * np.abs(Lref).max()
    now = (buf if where == "host" else buf.cpu().numpy()).reshape(lt * lt, mb * mb)
    before = user.reshape(lt * lt, mb * mb)
    for J in range(lt):
        for I in range(lt):
            inside = oi <= I < oi + vt and oj <= J < oj + vt
            t = now[I + J * lt]
            if not inside:
                assert np.array_equal(t, before[I + J * lt]), (I, J)
            elif I - oi >= J - oj:  # a lower tile of the view: holds the factor
                ref = Lref[(I - oi) * mb:(I - oi + 1) * mb, (J - oj) * mb:(J - oj + 1) * mb]
                got = t.reshape((mb, mb), order="F")
                got = np.tril(got) if I - oi == J - oj else got
                assert np.abs(got - ref).max() <= 1e-12 * np.abs(Lref).max(), (I, J)
    ch.CHAMELEON_Desc_Destroy(d)


@pytest.mark.parametrize("where", ["host", "device"])
def test_sub_matrix_view_with_unaligned_offsets(cham, orc, where):
    """v3_script_cholesky_x_arg_gpt.c:141-142, 186-212 accept any 0 <= i < lm: a view that starts INSIDE a tile of the user's
    matrix and ends inside another (round 4).  The view's entries are mirrored through an image tiled on its own; every
    entry of the user's buffer outside the view stays bit for bit what it was."""
    import torch

    ch = cham
    mb, lt, i0, j0, m = 256, 5, 100, 100, 700  # rows / columns 100 .. 799 of a 1280 x 1280 matrix in 256-tiles
    lm = lt * mb
    rng = np.random.default_rng(11)
    user = rng.standard_normal(lt * lt * mb * mb)

    def to_lapack(flat):  # the user's tile layout -> an lm x lm array
        t = flat.reshape(lt, lt, mb, mb)  # [J][I][jj][ii]
        return t.transpose(1, 3, 0, 2).reshape(lm, lm)

    A = orc.plgsy_matrix(m, float(m), 3)
    buf = user.copy() if where == "host" else torch.from_numpy(user.copy()).cuda()
    d = ch.CHAMELEON_Desc_Create(buf, ch.ChamRealDouble, mb, mb, mb * mb, lm, lm, i0, j0, m, m, 1, 1)
    d.from_lapack(A)
    mid = to_lapack(buf if where == "host" else buf.cpu().numpy())
    assert np.array_equal(mid[i0:i0 + m, j0:j0 + m], A)  # the view's entries landed where the offsets say
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    L = np.tril(d.to_lapack())
    Lref = np.linalg.cholesky(A)
    assert np.abs(L - Lref).max() <= 1e-12 * np.abs(Lref).max()
    after = to_lapack(buf if where == "host" else buf.cpu().numpy())
    before = to_lapack(user)
    outside = np.ones((lm, lm), dtype=bool)
    outside[i0:i0 + m, j0:j0 + m] = False
    assert np.array_equal(after[outside], before[outside])
    assert np.abs(np.tril(after[i0:i0 + m, j0:j0 + m]) - Lref).max() <= 1e-12 * np.abs(Lref).max()
    assert np.array_equal(np.triu(after[i0:i0 + m, j0:j0 + m], 1), np.triu(A, 1))  # strict upper triangle of the view untouched
    ch.CHAMELEON_Desc_Destroy(d)
    # a user matrix that is not made of whole tiles stays refused
    with pytest.raises(ch.CholmiError, match="whole square tiles"):
        ch.CHAMELEON_Desc_Create(buf, ch.ChamRealDouble, mb, mb, mb * mb, lm - 10, lm - 10, 100, 0, m, m, 1, 1)


def test_batches_on_the_two_streams_are_ordered_by_what_they_read(cham):
    """The executor behind chol_tile_batch (api.hip: TaskExec): URGENT updates, TRSM and POTRF batches run on the chain
    stream, other updates on the bulk stream, and a batch that reads a tile written on the other stream waits for exactly
    that batch.  A ladder of asynchronous batches that alternates between the streams, each consuming the tile the one
    before produced (long bulk batches ahead of short urgent ones, so that a missing wait would read a tile not yet
    written), against numpy; the statistics say that both streams were used and that events ordered them."""
    import torch

    from dense_linear_app_amd._lib import lib

    ch, L = cham, lib()
    B, steps, fill = 512, 6, 24
    rng = np.random.default_rng(11)

    def dev(a):
        return torch.from_numpy(np.asfortranarray(a).ravel(order="F").copy()).cuda()

    def host(t):
        return t.cpu().numpy().reshape((B, B), order="F")

    def ptrs(ts):
        return (C.c_void_p * len(ts))(*[int(t.data_ptr()) if t is not None else None for t in ts])

    C0 = rng.standard_normal((B, B))
    As = [rng.standard_normal((B, B)) / 8 for _ in range(steps)]
    Bs = [rng.standard_normal((B, B)) / 8 for _ in range(steps)]
    dA, dB = [dev(a) for a in As], [dev(b) for b in Bs]
    filler_c = [dev(rng.standard_normal((B, B))) for _ in range(fill)]
    torch.cuda.synchronize()
    st0 = (C.c_longlong * 4)()
    assert L.chol_batch_stats(st0) == 0
    cur, ref = dev(C0), C0.copy()
    keep = [cur]
    for s in range(steps):
        urgent = s % 2 == 1
        # the link of the ladder first in a bulk batch, behind it `fill` independent tasks: the next (urgent) batch must wait for all of it
        n = 1 if urgent else 1 + fill
        outs = [torch.empty(B * B, dtype=torch.float64, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        cin = [cur] + ([] if urgent else filler_c)
        rc = L.chol_tile_batch(4, ch.ChamRealDouble, B, n, ptrs(cin), ptrs([dA[s]] * n), ptrs([dB[s]] * n), ptrs(outs), None, 3 if urgent else 1)
        assert rc == 0, L.chol_last_error()
        ref = ref - As[s] @ Bs[s].T
        cur = outs[0]
        keep.append(outs)
    assert L.chol_sync() == 0
    assert np.abs(host(cur) - ref).max() <= 1e-12 * B * np.abs(ref).max()
    st1 = (C.c_longlong * 4)()
    assert L.chol_batch_stats(st1) == 0
    chain, bulk, waits = st1[0] - st0[0], st1[1] - st0[1], st1[2] - st0[2]
    assert chain == steps // 2 and bulk == steps - steps // 2 and waits >= steps - 2, (chain, bulk, waits)
    assert st1[3] == 0  # chol_sync forgot every tile
