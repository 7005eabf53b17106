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
