"""The distributed driver on real kernels: HipEngine + libcholmi wave kernels with P x Q > 1.
Only one GPU is available to the tests, so the ranks share cuda:0 and the broadcasts go
through gloo (RCCL needs one GPU per rank); ownership, local indexing, panel addressing and
stream ordering are exactly the multi-GPU code path."""
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


def _worker(rank, world, port, N, B, dtype, q, mode):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dense_linear_app_amd import distributed as dd

    P, Q = dd.grid_for(world)
    eng = dd.HipEngine(N, B, P, Q, rank, dtype, device=0)
    eng.generate(float(N), 42)
    chol = dd.BlockCyclicCholesky(eng, dist, lookahead=True, panel_mode=mode)
    info = chol.factorize()
    tiles = {}
    for I in range(eng.nt):
        for J in range(I + 1):
            if dd.owner_of(I, J, P, Q) == rank:
                tiles[(I, J)] = eng.download_tile(I, J)
    q.put((rank, info, tiles))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(1, "bcast"), (2, "bcast"), (4, "bcast"), (2, "allgather"), (4, "allgather")])
def test_hip_engine_block_cyclic(world, mode, orc):
    import torch.multiprocessing as mp

    N, B = 2304, 256  # 9 tiles per side: ragged parts and chunks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, B, "f64", q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = orc.tile_to_lapack(T, N, B)
    scale = np.abs(np.tril(Lref)).max()
    seen = 0
    for rank, info, tiles in got:
        assert info == 0
        for (I, J), t in tiles.items():
            ref = Lref[I * B:(I + 1) * B, J * B:(J + 1) * B]
            d = np.tril(t) - np.tril(ref) if I == J else t - ref
            assert np.abs(d).max() / scale <= 1e-12, (rank, I, J)
            seen += 1
    assert seen == (N // B) * (N // B + 1) // 2


def _worker_cabi(rank, world, port, N, B, q, bad):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dense_linear_app_amd import distributed as dd

    P, Q = dd.grid_for(world)
    eng = dd.HipEngine(N, B, P, Q, rank, "f64", device=0)
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

    full = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1) if rank == 0 else None
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


@pytest.mark.parametrize("world,bad", [(2, None), (4, None), (4, 1300)])
def test_potrf_tile_on_a_pxq_descriptor(world, bad, orc):
    """chol_potrf_tile works for a full P x Q descriptor.  The ranks share the one test GPU
    and the transport table is filled with gloo point-to-point calls (RCCL wants one GPU per rank);
    kernels, streams, ownership, addressing and the exchange pattern are the multi-GPU path."""
    import torch.multiprocessing as mp

    N, B = 2304, 256
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cabi, args=(r, world, port, N, B, q, bad)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
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
        assert (stats["residual"] <= 1e-13) if rank == 0 else stats["residual"] is None
        for (I, J), t in tiles.items():
            ref = Lref[I * B:(I + 1) * B, J * B:(J + 1) * B]
            d = np.tril(t) - np.tril(ref) if I == J else t - ref
            assert np.abs(d).max() / scale <= 1e-12, (rank, I, J)
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


def test_rccl_transport_loads_and_builds_a_communicator(cham):
    """One GPU cannot host two RCCL ranks, but the transport's bootstrap can run: librccl is found and
    resolved at run time, rank 0 creates an id, a one-rank communicator is built and torn down."""
    import ctypes as C

    from dense_linear_app_amd._lib import lib

    L = lib()
    ident = (C.c_char * 128)()
    assert L.chol_transport_rccl_unique_id(ident) == 0, L.chol_last_error()
    assert any(bytes(ident))
    assert L.chol_transport_rccl_init(ident, 0, 1) == 0, L.chol_last_error()
    assert L.chol_transport_rccl_init(ident, 0, 1) < 0  # one communicator per process
    assert L.chol_transport_rccl_finalize() == 0
