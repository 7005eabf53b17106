"""2D block-cyclic tiled Cholesky over several MI355X, one process per GPU.

Partitioning is Chameleon's own descriptor rule (p x q process grid, always 1 x 1 in the
reference: worker_distrib.cpp:77, v6_test.c:26-27): tile (I,J) lives on rank
(I mod P)*Q + (J mod Q); each rank stores its tiles packed over (I/P, J/Q).  Owner
computes.  The wave DAG is the reference client's (client_distrib.cpp:506-565); what
moves between GPUs per wave k is

  1. L(k,k): from its owner to the other ranks of process column k mod Q (they hold the
     rest of panel k and need it for their TRSMs);
  2. panel k, i.e. the tiles L(i,k), i > k: every one of the P ranks of that process column
     broadcasts its part (contiguous in its local storage, so no packing) to all ranks;
     each rank then runs its local SYRK/GEMM updates from the replicated panel.

The collectives are torch.distributed broadcasts (backend "nccl" = RCCL over xGMI on the
GPUs, "gloo" in the CPU tests).  There is no all-reduce anywhere.

Schedule (three streams per rank, receive buffers double-buffered by wave parity):

  main   U1(k) = update of columns k+1, k+2 by panel k;  U2(k) = the columns beyond
  side   TRSM(k+1), then the broadcasts of panel k+1 -- first its head tile L(k+2,k+1) alone,
         then the parts -- while U2(k) runs (one wave of lookahead)
  early  as soon as the head tile L(k+1,k) is in: the owner of (k+1,k+1) applies that one SYRK,
         factors the tile and broadcasts it down its process column -- while the rest of panel k
         is still on the wire.  POTRF and the L(k,k) broadcast are thereby off the per-wave
         critical path, which is  panel broadcast -> U1 -> TRSM -> next panel broadcast.

The wave logic is written against a small `engine` interface so that the CPU tests can
drive it with world_size 2 on gloo; the product engine is `HipEngine` (libcholmi.so,
no fallback).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np


def grid_for(nranks: int) -> Tuple[int, int]:
    """1x1, 1x2, 2x2, 2x4 for 1/2/4/8 GPUs; P <= Q in general."""
    p = int(math.isqrt(nranks))
    while nranks % p:
        p -= 1
    return p, nranks // p


def owner_of(I: int, J: int, P: int, Q: int) -> int:
    return (I % P) * Q + (J % Q)


def first_local_row_above(k: int, prow: int, P: int) -> int:
    """Local index of the first tile row i > k with i mod P == prow."""
    return (k + P - prow) // P


class HipEngine:
    """This rank's tiles in HBM + the wave-level kernels of libcholmi.so."""

    def __init__(self, N: int, B: int, P: int, Q: int, rank: int, dtype: str = "f64", device: Optional[int] = None):
        import torch

        from . import chameleon as ch
        from ._lib import check, lib

        self.torch, self.ch, self._lib, self._check = torch, ch, lib(), check
        assert N % B == 0
        self.N, self.B, self.P, self.Q, self.rank = N, B, P, Q, rank
        self.nt = N // B
        self.prow, self.pcol = rank // Q, rank % Q
        self.tdtype = torch.float64 if dtype == "f64" else torch.float32
        self.cdtype = ch.ChamRealDouble if dtype == "f64" else ch.ChamRealFloat
        # ONE device index for torch (tile storage, collectives) and for libcholmi (kernels):
        # the argument, else $LOCAL_RANK (what chol_init itself would pick), else cuda:0
        if device is None:
            import os

            device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(device)
        ch.set_device(device)
        ch.set_rank(rank, P * Q)
        ch.CHAMELEON_Init(1, 1)
        self.dev = torch.device("cuda", device)
        self.lmt = max(0, (self.nt - self.prow + P - 1) // P)
        self.lnt = max(0, (self.nt - self.pcol + Q - 1) // Q)
        self.bsiz = B * B
        self.store = torch.empty(max(1, self.lmt * self.lnt) * self.bsiz, dtype=self.tdtype, device=self.dev)
        self.desc = ch.CHAMELEON_Desc_Create(self.store, self.cdtype, B, B, self.bsiz, N, N, 0, 0, N, N, P, Q)
        assert self.desc.local_tiles() == (self.lmt, self.lnt)
        assert self.store.device.index == device

    # -- storage
    def empty_tiles(self, n: int):
        return self.torch.empty(max(1, n) * self.bsiz, dtype=self.tdtype, device=self.dev)

    def tiles_view(self, il: int, jl: int, count: int = 1):
        off = (il + jl * self.lmt) * self.bsiz
        return self.store[off:off + count * self.bsiz]

    def generate(self, bump: float, seed: int) -> None:
        self.ch.CHAMELEON_dplgsy_Tile(bump, self.ch.ChamLower, self.desc, seed)

    def upload_tile(self, I: int, J: int, tile: np.ndarray) -> None:
        self.desc.upload_tile(I, J, tile)

    def download_tile(self, I: int, J: int) -> np.ndarray:
        return self.desc.download_tile(I, J)

    # -- streams
    def new_stream(self):
        return self.torch.cuda.Stream(device=self.dev, priority=-1)

    def main_stream(self):
        # a stream of our own: torch's default stream is the HIP null stream, which would
        # serialise against every other stream of the process
        if getattr(self, "_main", None) is None:
            self._main = self.torch.cuda.Stream(device=self.dev)
        return self._main

    def stream_ctx(self, s):
        return self.torch.cuda.stream(s)

    def wait(self, waiter, waited) -> None:
        waiter.wait_stream(waited)

    def record(self, stream):
        ev = self.torch.cuda.Event()
        ev.record(stream)
        return ev

    def wait_event(self, stream, ev) -> None:
        stream.wait_event(ev)

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.dev)

    # -- wave kernels (asynchronous on stream s)
    def _sp(self, s):
        return C.c_void_p(s.cuda_stream)

    def potrf(self, k: int, lkk, s) -> None:
        self._check("chol_wave_potrf", self._lib.chol_wave_potrf(self.desc.handle, k, lkk.data_ptr(), self._sp(s)))

    def winv_elems(self) -> int:
        return int(self._lib.chol_wave_winv_bytes(self.desc.handle)) // self.store.element_size()

    def export_winv(self, dst, s) -> None:
        self._check("chol_wave_export_winv", self._lib.chol_wave_export_winv(self.desc.handle, dst.data_ptr(), self._sp(s)))

    def import_winv(self, src, s) -> None:
        self._check("chol_wave_import_winv", self._lib.chol_wave_import_winv(self.desc.handle, src.data_ptr(), self._sp(s)))

    def invert_diag(self, lkk, s) -> None:
        self._check("chol_wave_invert_diag",
                    self._lib.chol_wave_invert_diag(self.desc.handle, lkk.data_ptr(), self._sp(s)))

    def trsm(self, k: int, lkk, s) -> None:
        self._check("chol_wave_trsm", self._lib.chol_wave_trsm(self.desc.handle, k, lkk.data_ptr(), self._sp(s)))

    def update(self, k: int, jlo: int, jhi: int, bases: Sequence, firsts: Sequence[int], s,
               skip_diag: bool = False) -> None:
        pb = (C.c_void_p * self.P)(*[b.data_ptr() for b in bases])
        pf = (C.c_int * self.P)(*firsts)
        self._check("chol_wave_update",
                    self._lib.chol_wave_update(self.desc.handle, k, jlo, jhi, pb, pf, int(skip_diag), self._sp(s)))

    def update_diag(self, k: int, j: int, bases: Sequence, firsts: Sequence[int], s) -> None:
        pb = (C.c_void_p * self.P)(*[b.data_ptr() for b in bases])
        pf = (C.c_int * self.P)(*firsts)
        self._check("chol_wave_update_diag",
                    self._lib.chol_wave_update_diag(self.desc.handle, k, j, pb, pf, self._sp(s)))

    def reset_info(self) -> None:
        self._check("chol_reset_info", self._lib.chol_reset_info())

    def info(self) -> int:
        v = C.c_int()
        self._check("chol_get_info", self._lib.chol_get_info(C.byref(v)))
        return v.value

    def potrf_tile(self) -> int:
        """CHAMELEON_dpotrf_Tile(ChamLower, desc) on the p x q descriptor: the whole distributed
        factorisation inside the library (needs an installed transport when p*q > 1)."""
        return self.ch.CHAMELEON_dpotrf_Tile(self.ch.ChamLower, self.desc)

    def destroy(self) -> None:
        self.ch.CHAMELEON_Desc_Destroy(self.desc)


# ------------------------------------------------------------------------------------------
# The distributed factorisation behind the C ABI: chol_potrf_tile on a p x q descriptor runs the
# wave loop in C++ (csrc/dist.hip) and moves tiles through a transport table (include/cholmi.h,
# chol_transport_t).  Real runs: the RCCL transport (install_rccl_transport: point-to-point
# ncclSend / ncclRecv in groups over xGMI).  Tests and one-GPU rehearsals: TorchTransport, the same
# table filled with torch.distributed point-to-point calls (gloo).
# ------------------------------------------------------------------------------------------
class _TransportTable(C.Structure):
    _fields_ = [("ctx", C.c_void_p),
                ("group_begin", C.CFUNCTYPE(C.c_int, C.c_void_p)),
                ("send", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)),
                ("recv", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)),
                ("group_end", C.CFUNCTYPE(C.c_int, C.c_void_p)),
                ("allreduce_max", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_longlong)))]


class _DevPtr:
    """Raw device memory as a __cuda_array_interface__ object (torch.as_tensor wraps it without a copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class TorchTransport:
    """chol_transport_t over torch.distributed point-to-point operations (any backend).

    device=None: the buffers are host memory (the CPU tests of the wave loop);
    device=n: HBM of cuda:n -- the stream the library names is drained before a group moves, since
    gloo knows nothing of HIP streams.  One group = all isend / irecv posted, then all waited for."""

    def __init__(self, dist, device: Optional[int] = None, group=None):
        import torch

        self.dist, self.torch, self.device, self.group = dist, torch, device, group
        self.ops: List = []
        self.keep: List = []
        self.nbytes = 0
        T = _TransportTable
        self.table = T(None, T._fields_[1][1](self._begin), T._fields_[2][1](self._send), T._fields_[3][1](self._recv),
                       T._fields_[4][1](self._end), T._fields_[5][1](self._allreduce_max))

    def _view(self, ptr: int, nbytes: int):
        if self.device is None:
            buf = (C.c_char * nbytes).from_address(ptr)
            return self.torch.frombuffer(buf, dtype=self.torch.uint8)
        return self.torch.as_tensor(_DevPtr(ptr, nbytes), device=f"cuda:{self.device}")

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import traceback

            traceback.print_exc()
            self.error = e
            return -103

    def _begin(self, ctx):
        self.ops, self.keep = [], []
        return 0

    def _send(self, ctx, buf, nbytes, peer, stream):
        def f():
            t = self._view(buf, nbytes)
            self.keep.append(t)
            self.ops.append(("s", t, peer))
            self.nbytes += nbytes
        return self._guard(f)

    def _recv(self, ctx, buf, nbytes, peer, stream):
        def f():
            t = self._view(buf, nbytes)
            self.keep.append(t)
            self.ops.append(("r", t, peer))
        return self._guard(f)

    def _end(self, ctx):
        def f():
            if self.device is not None:
                self.torch.cuda.synchronize(self.device)
            if not self.ops:
                return
            if self.dist.get_backend(self.group) == "nccl":
                # RCCL through torch: the sends and receives of a group must progress together
                ops = [self.dist.P2POp(self.dist.isend if kind == "s" else self.dist.irecv, t, peer, self.group)
                       for kind, t, peer in self.ops]
                reqs = self.dist.batch_isend_irecv(ops)
            else:
                reqs = [self.dist.isend(t, peer, group=self.group) if kind == "s" else
                        self.dist.irecv(t, peer, group=self.group) for kind, t, peer in self.ops]
            for r in reqs:
                r.wait()
            if self.device is not None:
                self.torch.cuda.synchronize(self.device)
            self.ops, self.keep = [], []
        return self._guard(f)

    def _allreduce_max(self, ctx, value):
        def f():
            t = self.torch.tensor([value[0]], dtype=self.torch.int64)
            if self.dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
            value[0] = int(t.item())
        return self._guard(f)

    def install(self) -> None:
        from ._lib import check, lib

        check("chol_set_transport", lib().chol_set_transport(C.byref(self.table)))


def install_rccl_transport(dist) -> None:
    """Build the library's own RCCL communicator: rank 0 creates the id, torch.distributed (any
    backend; only its object broadcast is used) shares it, every rank joins."""
    from ._lib import check, lib

    L = lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = (C.c_char * 128)()
    box = [None]
    if rank == 0:  # a failure here must reach every rank, or they wait in the broadcast for ever
        rc = L.chol_transport_rccl_unique_id(buf)
        box = [bytes(buf) if rc == 0 else L.chol_last_error().decode(errors="replace")]
    dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], bytes):
        raise RuntimeError(f"chol_transport_rccl_unique_id failed on rank 0: {box[0]}")
    ident = (C.c_char * 128).from_buffer_copy(box[0])
    check("chol_transport_rccl_init", L.chol_transport_rccl_init(ident, rank, world))


def gather_lower(src_desc, dst_desc, root: int = 0) -> None:
    """Every rank: send the lower tiles of the p x q descriptor to `root`'s single-process descriptor
    (dst_desc is None elsewhere)."""
    from ._lib import check, lib

    check("chol_dist_gather_lower",
          lib().chol_dist_gather_lower(src_desc.handle, dst_desc.handle if dst_desc is not None else None, root))


def dist_last_stats() -> dict:
    from ._lib import lib

    us = C.c_double()
    a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
    lib().chol_dist_last_stats(C.byref(us), C.byref(a), C.byref(b), C.byref(c))
    return {"issue_us_per_wave": us.value, "sends": a.value, "recvs": b.value, "bytes_sent": c.value}


class BlockCyclicCholesky:
    """The distributed wave loop.  `dist` is torch.distributed (already initialised).

    panel_mode "bcast": each of the P owners of panel k broadcasts its part to all ranks.
    panel_mode "allgather": each owner scatters its part over its process row (Q chunks over Q
    distinct links), then ONE world all-gather replicates the panel -- every link carries 1/8
    of the panel instead of a ring carrying all of it.  Same result, same panel addressing.
    """

    def __init__(self, engine, dist, lookahead: bool = True, panel_mode: Optional[str] = None):
        import os

        self.e, self.dist, self.lookahead = engine, dist, lookahead
        self.panel_mode = panel_mode or os.environ.get("CHOLMI_PANEL_MODE", "bcast")
        assert self.panel_mode in ("bcast", "allgather")
        e = engine
        self.world = dist.get_world_size()
        assert self.world == e.P * e.Q and dist.get_rank() == e.rank
        # one group per process column (for L(k,k)); every rank must create every group
        self.col_groups = []
        for qc in range(e.Q):
            ranks = [pr * e.Q + qc for pr in range(e.P)]
            self.col_groups.append(dist.new_group(ranks=ranks) if e.P > 1 else None)
        maxpart = (e.nt + e.P - 1) // e.P
        # L(k,k) and, right behind it, the inverses of its 128-blocks: one broadcast per wave
        self.nwinv = e.winv_elems() if hasattr(e, "winv_elems") else 0
        self.lkk_buf = [e.empty_tiles(2)[:e.bsiz + self.nwinv] for _ in range(2)]
        self.head_buf = [e.empty_tiles(1) for _ in range(2)]  # L(k+1,k), sent ahead of the parts
        self._lkk = [None, None]
        if self.panel_mode == "bcast":
            # receive buffers: [parity][process row]
            self.pbuf = [[e.empty_tiles(maxpart) for _ in range(e.P)] for _ in range(2)]
        else:
            self.row_groups = []
            for pr in range(e.P):
                ranks = [pr * e.Q + qc for qc in range(e.Q)]
                self.row_groups.append(dist.new_group(ranks=ranks) if e.Q > 1 else None)
            self.chunkmax = (maxpart + e.Q - 1) // e.Q
            self.gbuf = [e.empty_tiles(self.world * self.chunkmax) for _ in range(2)]
            self.cbuf = [e.empty_tiles(self.chunkmax) for _ in range(2)]
            self.tail = [e.empty_tiles(self.chunkmax) for _ in range(2)]

    def warm_up(self) -> None:
        """Create every communicator this object will use (RCCL builds them lazily on first use)
        outside any timed region: one tiny collective per group."""
        e, dist = self.e, self.dist
        if self.world == 1:
            return
        t = e.empty_tiles(1)[:8]
        dist.broadcast(t, src=0)
        if e.P > 1:
            dist.broadcast(t, src=e.pcol, group=self.col_groups[e.pcol])
        if self.panel_mode == "allgather":
            if e.Q > 1:
                src = e.prow * e.Q
                dist.scatter(t, [t.clone() for _ in range(e.Q)] if e.rank == src else None, src=src,
                             group=self.row_groups[e.prow])
            out = e.empty_tiles(1)[:8 * self.world]
            try:
                dist.all_gather_into_tensor(out, t)
            except (RuntimeError, NotImplementedError, AttributeError):
                dist.all_gather([out[r * 8:(r + 1) * 8] for r in range(self.world)], t)
        e.synchronize()

    # -- L(k,k): POTRF on its owner, broadcast (with its block inverses) down the process column
    def _diag(self, k: int, s) -> None:
        e, dist = self.e, self.dist
        P, Q = e.P, e.Q
        pr, pc = k % P, k % Q
        par = k & 1
        if e.pcol != pc:
            return
        buf = self.lkk_buf[par]
        last = k + 1 >= e.nt
        if e.prow == pr:
            lkk = e.tiles_view(k // P, k // Q)
            e.potrf(k, lkk, s)
            if P > 1 and not last:
                # ship the factored tile together with the inverses of its diagonal blocks
                buf[:e.bsiz].copy_(lkk)
                if self.nwinv:
                    e.export_winv(buf[e.bsiz:], s)
        else:
            lkk = buf[:e.bsiz]
        if P > 1 and not last:
            dist.broadcast(buf, src=pr * Q + pc, group=self.col_groups[pc])
            if e.prow != pr:
                if self.nwinv:
                    e.import_winv(buf[e.bsiz:], s)
                else:
                    e.invert_diag(lkk, s)
        self._lkk[par] = lkk

    # -- TRSM of the local tiles of panel k, then its head tile L(k+1,k) ahead of everything else.
    # Returns (tile holding L(k+1,k) on this rank, event recorded once it is there).
    def _trsm_and_head(self, k: int, s, send_head: bool):
        e, dist = self.e, self.dist
        P, Q = e.P, e.Q
        if e.pcol == k % Q:
            e.trsm(k, self._lkk[k & 1], s)
        if not send_head or k + 1 >= e.nt:
            return None, None
        src = ((k + 1) % P) * Q + k % Q
        if e.rank == src:
            head = e.tiles_view((k + 1) // P, k // Q)
        else:
            head = self.head_buf[k & 1]
        if self.world > 1:
            dist.broadcast(head, src=src)
        return head, e.record(s)

    def _part(self, k: int, p2: int):
        """(first local row, tile count) of the part of panel k owned by process row p2."""
        e = self.e
        il0 = first_local_row_above(k, p2, e.P)
        return il0, max(0, (e.nt - p2 + e.P - 1) // e.P - il0)

    # -- one panel, mode "bcast"
    def _panel_bcast(self, k: int, s):
        e, dist = self.e, self.dist
        P, Q = e.P, e.Q
        pc, par = k % Q, k & 1
        bases, firsts = [], []
        for p2 in range(P):
            il0, cnt = self._part(k, p2)
            src = p2 * Q + pc
            if e.rank == src:
                buf = e.tiles_view(il0, k // Q, cnt) if cnt > 0 else self.pbuf[par][p2][:0]
            else:
                buf = self.pbuf[par][p2][:cnt * e.bsiz]
            if cnt > 0 and self.world > 1:
                dist.broadcast(buf, src=src)
            bases.append(buf if cnt > 0 else self.pbuf[par][p2])
            firsts.append(il0)
        return bases, firsts

    # -- one panel, mode "allgather"
    def _panel_allgather(self, k: int, s):
        e, dist = self.e, self.dist
        P, Q, bs = e.P, e.Q, e.bsiz
        pc, par = k % Q, k & 1
        parts = [self._part(k, p2) for p2 in range(P)]
        chunk = max((cnt + Q - 1) // Q for _, cnt in parts)
        gbuf, cbuf, tail = self.gbuf[par], self.cbuf[par], self.tail[par]
        firsts = [il0 for il0, _ in parts]
        bases = [gbuf[p2 * Q * chunk * bs:] if chunk > 0 else gbuf for p2 in range(P)]
        if chunk == 0:
            return bases, firsts
        # step 1: the owner of my process row's part scatters it over the row
        il0, cnt = parts[e.prow]
        src = e.prow * Q + pc
        mine = cbuf[:chunk * bs]
        views = None
        if e.rank == src:
            part = e.tiles_view(il0, k // Q, cnt) if cnt > 0 else tail[:0]
            views = []
            for q in range(Q):
                lo, hi = min(cnt, q * chunk), min(cnt, (q + 1) * chunk)
                if hi - lo == chunk:
                    views.append(part[lo * bs:hi * bs])
                else:
                    if hi > lo:  # the one ragged chunk: stage it so the send stays in bounds
                        tail[:(hi - lo) * bs].copy_(part[lo * bs:hi * bs])
                    views.append(tail[:chunk * bs])
        if Q > 1:
            dist.scatter(mine, views, src=src, group=self.row_groups[e.prow])
        else:
            mine.copy_(views[0])
        # step 2: one world all-gather replicates the panel; rank r's chunk lands at r*chunk
        out = gbuf[:self.world * chunk * bs]
        if self.world > 1:
            try:
                dist.all_gather_into_tensor(out, mine)
            except (RuntimeError, NotImplementedError, AttributeError):
                dist.all_gather([out[r * chunk * bs:(r + 1) * chunk * bs] for r in range(self.world)], mine)
        else:
            out.copy_(mine)
        return bases, firsts

    def _panel(self, k: int, s, send_head: bool = True):
        """TRSM(k) and the replication of panel k, on stream s.  -> (bases, firsts, head, head event)"""
        head, ev_head = self._trsm_and_head(k, s, send_head)
        bases, firsts = self._panel_bcast(k, s) if self.panel_mode == "bcast" else self._panel_allgather(k, s)
        return bases, firsts, head, ev_head

    def factorize(self) -> int:
        """In place on the engine's tiles.  Returns LAPACK info (max over ranks)."""
        e = self.e
        nt, P, Q = e.nt, e.P, e.Q
        main = e.main_stream()
        e.reset_info()
        if not self.lookahead:
            # the plain wave order on one stream (C2:506-565)
            with e.stream_ctx(main):
                for k in range(nt):
                    self._diag(k, main)
                    if k + 1 < nt:
                        bases, firsts, _, _ = self._panel(k, main, send_head=False)
                        e.update(k, k + 1, nt, bases, firsts, main)
        else:
            side, early = self._streams()
            e.wait(side, main)
            e.wait(early, main)
            with e.stream_ctx(early):
                self._diag(0, early)
            e.wait(side, early)
            panel = None
            if nt > 1:
                with e.stream_ctx(side):
                    panel = self._panel(0, side)
            ev_u1 = None
            for k in range(nt - 1):
                bases, firsts, head, ev_head = panel
                # early: the diagonal tile of the next wave, as soon as the head tile L(k+1,k) is in
                e.wait_event(early, ev_head)
                if ev_u1 is not None:
                    e.wait_event(early, ev_u1)  # (k+1,k+1) carries every update up to wave k-1
                with e.stream_ctx(early):
                    if owner_of(k + 1, k + 1, P, Q) == e.rank:
                        hb = [head] * P
                        hf = [0] * P
                        hf[(k + 1) % P] = (k + 1) // P
                        e.update_diag(k, k + 1, hb, hf, early)
                    self._diag(k + 1, early)
                ev_diag = e.record(early)
                # main: panel k complete (and received); columns k+1 and k+2 first
                e.wait(main, side)
                e.update(k, k + 1, k + 3, bases, firsts, main, skip_diag=True)
                ev_u1 = e.record(main)
                if k + 2 < nt:
                    e.wait_event(side, ev_u1)
                    e.wait_event(side, ev_diag)
                    with e.stream_ctx(side):
                        panel = self._panel(k + 1, side)
                e.update(k, k + 3, nt, bases, firsts, main)
            e.wait(main, side)
            e.wait(main, early)
        e.synchronize()
        info = e.info()
        if self.world > 1:
            import torch

            # the smallest positive info wins: MAX-reduce (2^40 - info), 0 = success
            v = (1 << 40) - info if info > 0 else 0
            t = torch.tensor([v], dtype=torch.int64, device=getattr(e, "dev", "cpu"))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            v = int(t.item())
            info = 0 if v == 0 else (1 << 40) - v
        return info

    def _streams(self):
        if getattr(self, "_side", None) is None:
            self._side, self._early = self.e.new_stream(), self.e.new_stream()
        return self._side, self._early


def run(N: int, B: int, dist, dtype: str = "f64", seed: int = 42, bump: Optional[float] = None,
        lookahead: bool = True, engine=None):
    """Generate the plgsy matrix (v6_test.c:46: bump = N) over the grid and factor it."""
    world, rank = dist.get_world_size(), dist.get_rank()
    P, Q = grid_for(world)
    if engine is None:
        engine = HipEngine(N, B, P, Q, rank, dtype)
    engine.generate(float(N) if bump is None else bump, seed)
    chol = BlockCyclicCholesky(engine, dist, lookahead)
    return chol, engine
