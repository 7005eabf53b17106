"""The multi-GPU wave logic (2D block-cyclic ownership, panel broadcasts, lookahead order)
under torch.distributed/gloo on the CPU, world sizes 2 (1x2) and 4 (2x2).  Tile arithmetic
comes from the oracle engine (tests only); the result must equal the single-process oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, B, lookahead, bad, q, mode="bcast"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dense_linear_app_amd import distributed as dd
    from oracle_engine import OracleEngine

    P, Q = dd.grid_for(world)
    eng = OracleEngine(N, B, P, Q, rank)
    eng.generate(float(N), 42)
    if bad is not None:  # make one pivot negative on its owner
        I = bad // B
        if dd.owner_of(I, I, P, Q) == rank:
            t = eng.download_tile(I, I)
            t[bad % B, bad % B] = -3.0
            eng.upload_tile(I, I, t)
    chol = dd.BlockCyclicCholesky(eng, dist, lookahead=lookahead, panel_mode=mode)
    chol.warm_up()
    info = chol.factorize()
    tiles = {}
    for I in range(eng.nt):
        for J in range(I + 1):
            if dd.owner_of(I, J, P, Q) == rank:
                tiles[(I, J)] = eng.download_tile(I, J)
    q.put((rank, info, tiles))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, N, B, lookahead=True, bad=None, mode="bcast"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, B, lookahead, bad, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,lookahead,mode", [(2, True, "bcast"), (2, False, "bcast"), (4, True, "bcast"),
                                                  (2, True, "allgather"), (4, True, "allgather"),
                                                  (4, False, "allgather"), (8, True, "bcast"),
                                                  (8, True, "allgather"), (6, True, "bcast")])
def test_block_cyclic_factorisation_matches_oracle(world, lookahead, mode):
    from oracle import oracle as orc

    N, B = 112, 16  # 7 tiles per side: ragged parts and ragged chunks
    got = _run(world, N, B, lookahead, mode=mode)
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = orc.tile_to_lapack(T, N, B)
    seen = 0
    for rank, info, tiles in got:
        assert info == 0
        for (I, J), t in tiles.items():
            ref = Lref[I * B:(I + 1) * B, J * B:(J + 1) * B]
            if I == J:
                assert np.abs(np.tril(t) - np.tril(ref)).max() <= 1e-13
            else:
                assert np.abs(t - ref).max() <= 1e-13
            seen += 1
    assert seen == (N // B) * (N // B + 1) // 2  # every lower tile owned exactly once


def test_info_is_agreed_by_all_ranks():
    got = _run(2, 64, 16, True, bad=37)
    assert [info for _, info, _ in got] == [38, 38]


def test_ownership_and_local_indexing():
    from dense_linear_app_amd import distributed as dd

    assert [dd.grid_for(n) for n in (1, 2, 4, 8)] == [(1, 1), (1, 2), (2, 2), (2, 4)]
    P, Q, nt = 2, 4, 11
    owners = {}
    for I in range(nt):
        for J in range(nt):
            owners.setdefault(dd.owner_of(I, J, P, Q), []).append((I, J))
    assert sorted(owners) == list(range(8))
    for k in range(nt):
        for pr in range(P):
            il0 = dd.first_local_row_above(k, pr, P)
            rows = [i for i in range(nt) if i % P == pr]
            assert [i for i in rows if i > k] == rows[il0:]
