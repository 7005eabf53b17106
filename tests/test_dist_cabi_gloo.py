"""The wave walker (csrc/walker.h, the code behind chol_potrf_tile on every whole tiled matrix, one GPU or a
p x q grid) driven on the CPU: the oracle's tile kernels plugged in through the library's test hook
(chol_dist_factorize_with), the tiles moved by a torch.distributed/gloo transport table -- ownership,
local indexing, matching of the point-to-point sends and receives along process rows and columns,
receive-buffer reuse, and the schedule's regimes (panels in pairs, near / far halves, plain waves: `mode`)
are exactly the product's code path; no compute of the product runs here.  World sizes 1, 2 (2x1), 3, 4 (2x2), 6 (3x2), 8 (4x2) -- the
default, tall grids -- and 1x2, 1x3, 2x3, 2x4, 4x1, 8x1 explicitly; the result must equal the single-process oracle and
every rank must report the same info."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
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


class OracleCEngine:
    """chol_test_engine_t over numpy storage and the oracle's dpotrf / dtrsm / dsyrk / dgemm."""

    class Table(C.Structure):
        _fields_ = [("ctx", C.c_void_p), ("store", C.c_void_p), ("esize", C.c_size_t),
                    ("alloc", C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)),
                    ("potrf", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)),
                    ("trsm", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)),
                    ("update", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_int), C.c_int)),
                    ("update_diag", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_int))),
                    ("info", C.CFUNCTYPE(C.c_int, C.c_void_p))]

    def __init__(self, orc, N, B, P, Q, rank):
        self.orc, self.N, self.B, self.P, self.Q, self.rank = orc, N, B, P, Q, rank
        self.nt = N // B
        self.prow, self.pcol = rank // Q, rank % Q
        self.lmt = max(0, (self.nt - self.prow + P - 1) // P)
        self.lnt = max(0, (self.nt - self.pcol + Q - 1) // Q)
        self.bsiz = B * B
        self.store = np.zeros(max(1, self.lmt * self.lnt) * self.bsiz)
        self.bufs = []
        self._info = 0
        T = self.Table
        f = dict(T._fields_)
        self.table = T(None, self.store.ctypes.data, 8, f["alloc"](self._alloc), f["potrf"](self._potrf),
                       f["trsm"](self._trsm), f["update"](self._update), f["update_diag"](self._update_diag),
                       f["info"](lambda ctx: self._info))

    def tile(self, il, jl):
        off = (il + jl * self.lmt) * self.bsiz
        return self.store[off:off + self.bsiz].reshape((self.B, self.B), order="F")

    def _at(self, ptr):
        return np.ctypeslib.as_array((C.c_double * self.bsiz).from_address(ptr)).reshape((self.B, self.B), order="F")

    def _ptile(self, bases, firsts, i):
        return self._at(bases[i % self.P] + (i // self.P - firsts[i % self.P]) * self.bsiz * 8)

    def generate(self, bump, seed):
        for jl in range(self.lnt):
            for il in range(self.lmt):
                I, J = il * self.P + self.prow, jl * self.Q + self.pcol
                t = self.tile(il, jl)
                for jj in range(self.B):
                    for ii in range(self.B):
                        t[ii, jj] = self.orc.plgsy_entry(bump, seed, self.N, I * self.B + ii, J * self.B + jj)

    def _alloc(self, ctx, nbytes):
        a = np.full(max(1, nbytes // 8), np.nan)  # NaN: a tile used without having been received shows up
        self.bufs.append(a)
        return a.ctypes.data

    def _potrf(self, ctx, k, lkk):
        t = self._at(lkk)
        out, info = self.orc.dpotrf(t)
        t[:, :] = out
        if info and not self._info:
            self._info = k * self.B + info
        return 0

    def _trsm(self, ctx, k, lkk):
        if k % self.Q != self.pcol:
            return 0
        L = self._at(lkk)
        for il in range((k + self.P - self.prow) // self.P, self.lmt):
            t = self.tile(il, k // self.Q)
            t[:, :] = self.orc.dtrsm(L, t)
        return 0

    def _update_diag(self, ctx, k, j, bases, firsts):
        assert j % self.Q == self.pcol and j % self.P == self.prow and k < j < self.nt
        Cm = self.tile(j // self.P, j // self.Q)
        Cm[:, :] = self.orc.dsyrk(self._ptile(bases, firsts, j), Cm)
        return 0

    def _update(self, ctx, k, jlo, jhi, bases, firsts, skip_diag):
        jlo = max(jlo, k + 1)
        for j in range(jlo, min(jhi, self.nt)):
            if j % self.Q != self.pcol:
                continue
            for i in range(j, self.nt):
                if i % self.P != self.prow or (skip_diag and i == j == jlo):
                    continue
                Ai, Aj = self._ptile(bases, firsts, i), self._ptile(bases, firsts, j)
                assert np.isfinite(Ai).all() and np.isfinite(Aj).all(), (self.rank, k, i, j)
                Cm = self.tile(i // self.P, j // self.Q)
                Cm[:, :] = self.orc.dsyrk(Ai, Cm) if i == j else self.orc.dgemm(Ai, Aj, Cm)
        return 0


def _worker(rank, world, port, N, B, mode, bad, q, grid):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dense_linear_app_amd import distributed as dd
    from dense_linear_app_amd._lib import lib
    from oracle import oracle as orc

    P, Q = grid if grid else dd.grid_for(world)
    eng = OracleCEngine(orc, N, B, P, Q, rank)
    eng.generate(float(N), 42)
    if bad is not None:
        I = bad // B
        if dd.owner_of(I, I, P, Q) == rank:
            eng.tile(I // P, I // Q)[bad % B, bad % B] = -3.0
    tr = dd.TorchTransport(dist, device=None)
    info = lib().chol_dist_factorize_with(C.byref(eng.table), C.byref(tr.table), N, B, P, Q, rank, int(mode))
    stats = dd.dist_last_stats()
    tiles = {}
    for I in range(eng.nt):
        for J in range(I + 1):
            if dd.owner_of(I, J, P, Q) == rank:
                tiles[(I, J)] = eng.tile(I // P, J // Q).copy()
    q.put((rank, info, tiles, stats))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, N, B, mode=1, bad=None, grid=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, B, mode, bad, q, grid)) for r in range(world)]
    for p in procs:
        p.daemon = True  # a rank that hangs must not keep the test runner from exiting
        p.start()
    try:
        got = [q.get(timeout=240) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    return got


@pytest.mark.parametrize("world,mode,grid", [(1, 0, None), (1, 1, None), (1, 2, None), (2, 1, None), (2, 0, None), (2, 2, None),
                                             (4, 1, None), (4, 0, None), (4, 2, None), (6, 1, None), (8, 1, None), (8, 2, None),
                                             (2, 1, (1, 2)), (3, 1, (3, 1)), (3, 2, (1, 3)), (6, 0, (2, 3)), (8, 1, (2, 4)),
                                             (8, 0, (8, 1)), (4, 1, (4, 1))])
def test_c_wave_loop_matches_oracle(world, mode, grid):
    from oracle import oracle as orc

    N, B = 176, 16  # 11 tiles per side: ragged parts, every residue class of (i mod p, j mod q), > 2 buffer rotations
    got = _run(world, N, B, mode, grid=grid)
    T = orc.plgsy_tiles(N // B, B, float(N), 42)
    assert orc.tiled_potrf(T, N // B, B) == 0
    Lref = orc.tile_to_lapack(T, N, B)
    seen = 0
    for rank, info, tiles, stats in got:
        assert info == 0, (rank, info)
        for (I, J), t in tiles.items():
            ref = Lref[I * B:(I + 1) * B, J * B:(J + 1) * B]
            d = np.tril(t) - np.tril(ref) if I == J else t - ref
            assert np.abs(d).max() <= 1e-13, (rank, I, J)
            seen += 1
    assert seen == (N // B) * (N // B + 1) // 2
    # every byte sent is received, and nobody gets the whole panel: with p x q ranks a rank receives about
    # (1/p + 1/q) of each panel, not (1 - 1/(p q)) of it
    sends = sum(s["sends"] for *_, s in got)
    recvs = sum(s["recvs"] for *_, s in got)
    assert sends == recvs
    if world == 8 and mode == 1:
        nt, tile = N // B, B * B * 8
        replicated = sum((nt - 1 - k) * tile * (world - 1) for k in range(nt))
        moved = sum(s["bytes_sent"] for *_, s in got)
        assert moved < 0.85 * replicated  # about (1/p + 1/q) of every panel per rank: 0.75 on 4 x 2 and 2 x 4


def test_c_wave_loop_info_is_agreed_by_all_ranks():
    got = _run(4, 64, 16, 1, bad=37)
    assert [info for _, info, _, _ in got] == [38, 38, 38, 38]


def test_distributed_potrf_without_transport_is_refused():
    """No GPU needed: the argument check of the test hook itself."""
    sys.path.insert(0, ROOT)
    from dense_linear_app_amd._lib import lib

    L = lib()
    assert L.chol_dist_factorize_with(None, None, 64, 16, 1, 2, 0, 1) < 0
    assert b"engine" in L.chol_last_error()
