"""The p x q walker on real kernels with P x Q > 1, on the one GPU a test box has:
  * chol_dist_rehearse: the ranks are THREADS of this process, each with its own streams and workspaces, tiles moved by
    stream-ordered device copies -- nothing ever synchronises the device, so the stream / event ordering and the rotation
    of the receive buffers are what is tested;
  * chol_potrf_tile on a p x q descriptor with the ranks as PROCESSES sharing cuda:0 and a gloo transport table (RCCL
    wants one GPU per rank): the C ABI path end to end, gather and residual included;
  * the RCCL transport itself: version check, two one-rank communicators, a tile to self on each channel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check_against_oracle(orc, full, N, B, dtype, tol):
    """The gathered factor (a 1 x 1 descriptor) against the oracle's tiled factorisation, entry by entry."""
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = np.tril(orc.tile_to_lapack(T, N, B))
    L = np.tril(full.to_lapack()).astype(np.float64)
    assert np.abs(L - Lref).max() / np.abs(Lref).max() <= tol


@pytest.mark.parametrize("P,Q,N,B,dtype", [(1, 1, 2304, 256, "f64"), (1, 2, 2304, 256, "f64"), (2, 1, 2304, 256, "f64"),
                                           (2, 2, 2304, 256, "f64"), (3, 1, 2304, 256, "f64"), (2, 3, 2304, 256, "f64"),
                                           (2, 4, 3328, 256, "f64"), (4, 2, 3328, 256, "f64"), (4, 1, 2304, 256, "f64"),
                                           (2, 2, 8192, 512, "f64"), (4, 2, 8192, 512, "f64"), (2, 4, 8192, 512, "f64"),
                                           (2, 2, 4096, 512, "f32"), (1, 3, 2304, 128, "f32")])
def test_pxq_rehearsal_on_one_gpu(P, Q, N, B, dtype, cham, orc):
    """The p x q walker with its real kernels, streams, events and receive-buffer rotation: the ranks are threads of
    this process on the one GPU, every send / receive is a stream-ordered device copy and nothing ever synchronises
    the device (chol_dist_rehearse) -- a missing event edge or a buffer reused too early gives wrong data here."""
    from dense_linear_app_amd import distributed as dd

    ch = cham
    info, full, ms = dd.rehearse(N, B, P, Q, dtype)
    assert info == 0
    res = ch.residual_plgsy(full, float(N), 42)
    assert res <= (1e-13 if dtype == "f64" else 5e-5)
    _check_against_oracle(orc, full, N, B, dtype, 1e-12 if dtype == "f64" else 1e-4)
    ch.CHAMELEON_Desc_Destroy(full)


@pytest.mark.parametrize("N,dtype,tol", [(65536, "f64", 1e-13)])  # (config 5, N=131072 fp32: scripts/rehearse_full_size.py f32 -- 20 s)
def test_baseline_configs_4_and_5_partitioned_for_8_gpus_at_full_size(N, dtype, tol, cham):
    """BASELINE configs 4 and 5 name a 2D block-cyclic partitioning over 8 GPUs.  There is one GPU here: the 4 x 2 walker of
    all eight ranks (threads, stream-ordered copies, no device synchronisation: chol_dist_rehearse) at the FULL size, the
    gathered factor checked by its residual; fp64 also against the one-GPU walker's factor of the same matrix, on the
    device, with the driver's own operations (dlacpy Lower, dgeadd, dlange Max).  scripts/rehearse_full_size.py does the
    same for 2 x 4 as well (profiles/r05_rehearse_full_size.txt)."""
    import torch

    from dense_linear_app_amd import distributed as dd

    ch = cham
    B = 1024
    info, full, ms = dd.rehearse(N, B, 4, 2, dtype)
    assert info == 0
    assert ch.residual_plgsy(full, float(N), 42) <= tol
    if dtype == "f64":
        mk = lambda: ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
        ref, t1, t2 = mk(), mk(), mk()
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, ref, 42)
        assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, ref) == 0
        for t in (t1, t2):
            ch.CHAMELEON_dgeadd_Tile(ch.ChamNoTrans, 0.0, t, 0.0, t)
        ch.CHAMELEON_dlacpy_Tile(ch.ChamLower, ref, t1)
        ch.CHAMELEON_dlacpy_Tile(ch.ChamLower, full, t2)
        scale = ch.CHAMELEON_dlange_Tile(ch.ChamMaxNorm, t1)
        ch.CHAMELEON_dgeadd_Tile(ch.ChamNoTrans, -1.0, t2, 1.0, t1)
        assert ch.CHAMELEON_dlange_Tile(ch.ChamMaxNorm, t1) / scale <= 1e-12
        for t in (ref, t1, t2):
            ch.CHAMELEON_Desc_Destroy(t)
    ch.CHAMELEON_Desc_Destroy(full)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("P,Q,N,B", [(2, 2, 2000, 256), (2, 3, 1500, 192)])
def test_pxq_rehearsal_ragged_order_and_odd_tiles(P, Q, N, B, cham, orc):
    """N not a multiple of the tile, tile not a multiple of 128, on a p x q grid: every rank keeps its padded image."""
    from dense_linear_app_amd import distributed as dd

    ch = cham
    info, full, _ = dd.rehearse(N, B, P, Q)
    assert info == 0
    A = orc.plgsy_matrix(N, float(N), 42)
    Lref, iref = orc.cholesky_lower_any(A, B)
    assert iref == 0
    L = np.tril(full.to_lapack())
    assert np.abs(L - Lref).max() / np.abs(Lref).max() <= 1e-12
    assert np.linalg.norm(L @ L.T - A) / np.linalg.norm(A) <= 1e-13
    ch.CHAMELEON_Desc_Destroy(full)


def test_one_by_one_through_the_rehearsal_is_the_walker(cham):
    """p = q = 1 is the same code path as chol_potrf_tile on an ordinary descriptor: bit-identical factors."""
    from dense_linear_app_amd import distributed as dd

    ch = cham
    N, B = 4096, 512
    info, full, _ = dd.rehearse(N, B, 1, 1)
    assert info == 0
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d) == 0
    assert np.array_equal(np.tril(full.to_lapack()), np.tril(d.to_lapack()))
    ch.CHAMELEON_Desc_Destroy(full)
    ch.CHAMELEON_Desc_Destroy(d)


def test_rehearsal_reports_a_bad_pivot_on_every_rank(cham):
    """info is agreed by all ranks (MAX-reduce of the smallest index): plgsy with a bump that is too small."""
    from dense_linear_app_amd import distributed as dd

    info, full, _ = dd.rehearse(2304, 256, 2, 2, bump=1.0)
    assert info > 0
    cham.CHAMELEON_Desc_Destroy(full)


def _worker_cabi(rank, world, port, N, B, q, bad, grid, dtype):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # Should a rank ever sit in a transport call again (round 3: one 300 s hang of the 3 x 1 case, no stack kept), its
    # Python stack goes to gpurun_out/ after 200 s -- before the parent's 300 s limit -- so that the blocked call is on file.
    import faulthandler

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    _stk = open(os.path.join(ROOT, "gpurun_out", f"pxq_rank{rank}_of_{world}_stack.txt"), "w")
    faulthandler.dump_traceback_later(200, file=_stk, exit=False)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dense_linear_app_amd import distributed as dd

    P, Q = grid if grid else dd.grid_for(world)
    eng = dd.HipEngine(N, B, P, Q, rank, dtype, device=0)
    eng.generate(float(N), 42)
    if bad is not None:
        I = bad // B
        if dd.owner_of(I, I, P, Q) == rank:
            t = eng.download_tile(I, I)
            t[bad % B, bad % B] = -3.0
            eng.upload_tile(I, I, t)
    tr = dd.TorchTransport(dist, device=0)
    tr.install()
    info = eng.potrf_tile()  # CHAMELEON_dpotrf_Tile(ChamLower, descriptor with p*q > 1): the C++ wave loop
    stats = dd.dist_last_stats()
    # the verification step bench.py takes after a multi-GPU run: the factor gathered on rank 0, residual there
    from dense_linear_app_amd import chameleon as ch

    cd = ch.ChamRealDouble if dtype == "f64" else ch.ChamRealFloat
    full = ch.CHAMELEON_Desc_Create(None, cd, B, B, B * B, N, N, 0, 0, N, N, 1, 1) if rank == 0 else None
    dd.gather_lower(eng.desc, full, 0)
    stats["residual"] = ch.residual_plgsy(full, float(N), 42) if (rank == 0 and bad is None) else None
    tiles = {}
    for I in range(eng.nt):
        for J in range(I + 1):
            if dd.owner_of(I, J, P, Q) == rank:
                tiles[(I, J)] = eng.download_tile(I, J)
    q.put((rank, info, tiles, stats))
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    _stk.close()
    os.unlink(_stk.name)  # (nothing to report)


@pytest.mark.parametrize("world,bad,grid,dtype", [(2, None, None, "f64"), (4, None, None, "f64"), (4, 1300, None, "f64"),
                                                  (2, None, (1, 2), "f64"), (3, None, (3, 1), "f64"), (4, None, None, "f32"),
                                                  (4, None, (4, 1), "f64")])
def test_potrf_tile_on_a_pxq_descriptor(world, bad, grid, dtype, orc):
    """chol_potrf_tile works for a full P x Q descriptor.  The ranks share the one test GPU
    and the transport table is filled with gloo point-to-point calls (RCCL wants one GPU per rank);
    kernels, streams, ownership, addressing and the exchange pattern are the multi-GPU path."""
    import torch.multiprocessing as mp

    N, B = 2304, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cabi, args=(r, world, port, N, B, q, bad, grid, dtype)) for r in range(world)]
    for p in procs:
        p.daemon = True  # a rank that hangs must not keep the test runner from exiting
        p.start()
    try:
        got = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    if bad is not None:
        assert [info for _, info, _, _ in got] == [bad + 1] * world
        return
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = orc.tile_to_lapack(T, N, B)
    scale = np.abs(np.tril(Lref)).max()
    seen = 0
    for rank, info, tiles, stats in got:
        assert info == 0
        assert stats["sends"] > 0 and stats["issue_us_per_wave"] > 0
        assert (stats["residual"] <= (1e-13 if dtype == "f64" else 5e-5)) if rank == 0 else stats["residual"] is None
        for (I, J), t in tiles.items():
            ref = Lref[I * B:(I + 1) * B, J * B:(J + 1) * B]
            d = np.tril(t) - np.tril(ref) if I == J else t - ref
            assert np.abs(d).max() / scale <= (1e-12 if dtype == "f64" else 1e-4), (rank, I, J)
            seen += 1
    assert seen == (N // B) * (N // B + 1) // 2


def test_potrf_tile_on_a_pxq_descriptor_needs_a_transport(cham):
    """Without a transport the distributed descriptor is refused, loudly."""
    from dense_linear_app_amd._lib import lib

    ch = cham
    L = lib()
    L.chol_set_transport(None)
    ch.set_rank(0, 2)
    try:
        d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, 256, 256, 256 * 256, 1024, 1024, 0, 0, 1024, 1024, 1, 2)
        with pytest.raises(ch.CholmiError, match="transport"):
            ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        ch.CHAMELEON_Desc_Destroy(d)
    finally:
        ch.set_rank(0, 1)


def test_rccl_transport_moves_tiles_to_self_on_both_channels(cham):
    """One GPU cannot host two RCCL ranks, but the transport's own entries can execute: librccl is found and
    resolved at run time and its version checked, rank 0 creates the id blob, one-rank communicators are built (one per
    channel), and an 8 MiB tile travels through ncclSend / ncclRecv to self inside ncclGroupStart / ncclGroupEnd
    on EACH channel, both groups in flight together on the walker's two communication streams; byte-compared."""
    import ctypes as C

    from dense_linear_app_amd import distributed as dd
    from dense_linear_app_amd._lib import lib

    L = lib()
    assert L.chol_transport_rccl_version() >= 20700
    ident = (C.c_char * dd.RCCL_ID_BYTES)()
    assert L.chol_transport_rccl_unique_id(ident) == 0, L.chol_last_error()
    assert any(bytes(ident)[:128]) and any(bytes(ident)[128:]) and bytes(ident)[:128] != bytes(ident)[128:]
    assert L.chol_transport_rccl_init(ident, 0, 1) == 0, L.chol_last_error()
    try:
        assert L.chol_transport_rccl_init(ident, 0, 1) < 0  # one set of communicators per process
        dd.transport_selftest(0, 8 << 20)
        dd.transport_selftest(0, 1024)
    finally:
        assert L.chol_transport_rccl_finalize() == 0
