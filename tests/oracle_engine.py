"""Test-only engine for dense_linear_app_amd.distributed.BlockCyclicCholesky: the same
storage layout and panel addressing as HipEngine, with the CPU oracle's tile kernels and
CPU tensors, so the distribution logic can run under gloo without a GPU."""
import contextlib

import numpy as np
import torch

from oracle import oracle as orc


class OracleEngine:
    def __init__(self, N, B, P, Q, rank):
        assert N % B == 0
        self.N, self.B, self.P, self.Q, self.rank = N, B, P, Q, rank
        self.nt = N // B
        self.prow, self.pcol = rank // Q, rank % Q
        self.lmt = max(0, (self.nt - self.prow + P - 1) // P)
        self.lnt = max(0, (self.nt - self.pcol + Q - 1) // Q)
        self.bsiz = B * B
        self.store = torch.zeros(max(1, self.lmt * self.lnt) * self.bsiz, dtype=torch.float64)
        self._info = 0
        self.dev = "cpu"

    def empty_tiles(self, n):
        return torch.zeros(max(1, n) * self.bsiz, dtype=torch.float64)

    def tiles_view(self, il, jl, count=1):
        off = (il + jl * self.lmt) * self.bsiz
        return self.store[off:off + count * self.bsiz]

    def _np_tile(self, t, idx=0):
        return t.numpy()[idx * self.bsiz:(idx + 1) * self.bsiz].reshape((self.B, self.B), order="F")

    def generate(self, bump, seed):
        for jl in range(self.lnt):
            for il in range(self.lmt):
                I, J = il * self.P + self.prow, jl * self.Q + self.pcol
                t = self._np_tile(self.tiles_view(il, jl))
                for jj in range(self.B):
                    for ii in range(self.B):
                        t[ii, jj] = orc.plgsy_entry(bump, seed, self.N, I * self.B + ii, J * self.B + jj)

    def upload_tile(self, I, J, tile):
        self._np_tile(self.tiles_view(I // self.P, J // self.Q))[:, :] = tile

    def download_tile(self, I, J):
        return self._np_tile(self.tiles_view(I // self.P, J // self.Q)).copy(order="F")

    # streams are no-ops on the CPU
    def new_stream(self):
        return object()

    def main_stream(self):
        return object()

    def stream_ctx(self, s):
        return contextlib.nullcontext()

    def wait(self, a, b):
        pass

    def record(self, s):
        return object()

    def wait_event(self, s, ev):
        pass

    def synchronize(self):
        pass

    def potrf(self, k, lkk, s):
        t = self._np_tile(lkk)
        out, info = orc.dpotrf(t)
        t[:, :] = out
        if info and not self._info:
            self._info = k * self.B + info

    def invert_diag(self, lkk, s):
        pass

    def trsm(self, k, lkk, s):
        if k % self.Q != self.pcol:
            return
        L = self._np_tile(lkk)
        il0 = (k + self.P - self.prow) // self.P
        for il in range(il0, self.lmt):
            t = self._np_tile(self.tiles_view(il, k // self.Q))
            t[:, :] = orc.dtrsm(L, t)

    def update_diag(self, k, j, bases, firsts, s):
        assert j % self.Q == self.pcol and j % self.P == self.prow and k < j < self.nt
        Aj = self._np_tile(bases[j % self.P], j // self.P - firsts[j % self.P])
        Cm = self._np_tile(self.tiles_view(j // self.P, j // self.Q))
        Cm[:, :] = orc.dsyrk(Aj, Cm)

    def update(self, k, jlo, jhi, bases, firsts, s, skip_diag=False):
        jlo = max(jlo, k + 1)
        for j in range(jlo, min(jhi, self.nt)):
            if j % self.Q != self.pcol:
                continue
            for i in range(j, self.nt):
                if i % self.P != self.prow or (skip_diag and i == j == jlo):
                    continue
                Ai = self._np_tile(bases[i % self.P], i // self.P - firsts[i % self.P])
                Aj = self._np_tile(bases[j % self.P], j // self.P - firsts[j % self.P])
                Cm = self._np_tile(self.tiles_view(i // self.P, j // self.Q))
                Cm[:, :] = orc.dsyrk(Ai, Cm) if i == j else orc.dgemm(Ai, Aj, Cm)

    def reset_info(self):
        self._info = 0

    def info(self):
        return self._info


class OracleTileBackend:
    """Test-only worker backend: the four tile routines on the CPU oracle."""

    def potrf(self, A, B):
        out, info = orc.dpotrf(A.reshape((B, B), order="F"))
        A[:] = out.ravel(order="F")
        return info

    def trsm(self, L, A, B):
        A[:] = orc.dtrsm(L.reshape((B, B), order="F"), A.reshape((B, B), order="F")).ravel(order="F")
        return 0

    def syrk(self, A, Cm, B):
        Cm[:] = orc.dsyrk(A.reshape((B, B), order="F"), Cm.reshape((B, B), order="F")).ravel(order="F")
        return 0

    def gemm(self, Ai, Aj, Cm, B):
        Cm[:] = orc.dgemm(Ai.reshape((B, B), order="F"), Aj.reshape((B, B), order="F"),
                          Cm.reshape((B, B), order="F")).ravel(order="F")
        return 0
