"""2D block-cyclic tiled Cholesky over several MI355X, one process per GPU: the Python side.

Partitioning is Chameleon's own descriptor rule (p x q process grid, always 1 x 1 in the
reference: worker_distrib.cpp:77, v6_test.c:26-27): tile (I,J) lives on rank
(I mod P)*Q + (J mod Q); each rank stores its tiles packed over (I/P, J/Q).  Owner computes.

The factorisation itself is ONE call, CHAMELEON_dpotrf_Tile on the p x q descriptor: the wave walker of
libcholmi (csrc/walker.h -- the same schedule as on one GPU) moves tiles through a transport table
(include/cholmi.h, chol_transport_t; two channels).  This module holds
  * HipEngine: this rank's tiles in HBM behind a p x q descriptor (generate, upload / download, potrf_tile);
  * install_rccl_transport: the library's own RCCL communicators (point-to-point ncclSend / ncclRecv in groups
    over xGMI), bootstrapped over any torch.distributed backend;
  * TorchTransport: the same table filled with torch.distributed point-to-point calls (gloo) for the CPU
    tests of the walker and for rehearsals where several ranks share the GPUs that exist;
  * rehearse: p*q ranks as threads of one process on ONE GPU with an asynchronous in-process transport
    (chol_dist_rehearse) -- the real kernels, streams and buffer reuse without any device synchronisation.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Tuple

import numpy as np


def grid_for(nranks: int) -> Tuple[int, int]:
    """The process grid for `nranks` GPUs: as square as possible, P >= Q -- 1x1, 2x1, 2x2, 4x2 for 1/2/4/8.
    Tall rather than wide (SURVEY 8e suggests 2x4): what bounds a wave once the update is short is the panel cycle
    TRSM -> exchange -> update of column k+1, whose critical transfer is the part of the panel one process row holds
    (nt/P tiles) over ONE xGMI link to the next process column; a tall grid halves that part and the local TRSM, and
    the fully connected links carry the extra column-operand messages side by side (scripts/grid_model.py:
    2x4 5.7x, 4x2 6.8x of the measured 1-GPU time at N=65536, tile 1024).  CHOLMI_GRID=PxQ overrides."""
    import os

    env = os.environ.get("CHOLMI_GRID")
    if env:
        p, q = (int(x) for x in env.lower().split("x"))
        if p * q != nranks:
            raise ValueError(f"CHOLMI_GRID={env} does not match {nranks} ranks")
        return p, q
    q = int(math.isqrt(nranks))
    while nranks % q:
        q -= 1
    return nranks // q, q


def candidate_grids(nranks: int) -> list:
    """The grids worth MEASURING for `nranks` GPUs: grid_for's and its transpose (4x2 and 2x4 for 8).  Which of the two is
    faster depends on the links (the tall grid halves the panel part that crosses ONE link per wave, the wide one the
    column-operand messages), and the model that prefers the tall one rests on an assumed bandwidth and an assumed
    per-group latency (scripts/grid_model.py); with communication free the wide grid is the faster
    (profiles/r04_rank_alone_grids.txt).  bench.py --gpus N therefore runs one warm-up factorisation on each and takes
    the faster (config.grid_probe).  CHOLMI_GRID pins one."""
    import os

    p, q = grid_for(nranks)
    if os.environ.get("CHOLMI_GRID") or p == q:
        return [(p, q)]
    return [(p, q), (q, p)]


def owner_of(I: int, J: int, P: int, Q: int) -> int:
    return (I % P) * Q + (J % Q)


def first_local_row_above(k: int, prow: int, P: int) -> int:
    """Local index of the first tile row i > k with i mod P == prow."""
    return (k + P - prow) // P


class HipEngine:
    """This rank's tiles in HBM behind a p x q descriptor of libcholmi.so."""

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
        # ONE device index for torch (tile storage) and for libcholmi (kernels):
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

    def generate(self, bump: float, seed: int) -> None:
        self.ch.CHAMELEON_dplgsy_Tile(bump, self.ch.ChamLower, self.desc, seed)

    def upload_tile(self, I: int, J: int, tile: np.ndarray) -> None:
        self.desc.upload_tile(I, J, tile)

    def download_tile(self, I: int, J: int) -> np.ndarray:
        return self.desc.download_tile(I, J)

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.dev)

    def potrf_tile(self) -> int:
        """CHAMELEON_dpotrf_Tile(ChamLower, desc) on the p x q descriptor: the whole distributed
        factorisation inside the library (needs an installed transport when p*q > 1)."""
        return self.ch.CHAMELEON_dpotrf_Tile(self.ch.ChamLower, self.desc)

    def destroy(self) -> None:
        self.ch.CHAMELEON_Desc_Destroy(self.desc)
        self.store = None


# ------------------------------------------------------------------------------------------
# The distributed factorisation behind the C ABI: chol_potrf_tile on a p x q descriptor runs the
# wave walker in C++ (csrc/walker.h, csrc/dist.hip) and moves tiles through a transport table (include/cholmi.h,
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
        if getattr(self, "trace", False):
            print(f"[transport {self.dist.get_rank()}] send {nbytes} B -> {peer}", flush=True)

        def f():
            t = self._view(buf, nbytes)
            self.keep.append(t)
            self.ops.append(("s", t, peer))
            self.nbytes += nbytes
        return self._guard(f)

    def _recv(self, ctx, buf, nbytes, peer, stream):
        if getattr(self, "trace", False):
            print(f"[transport {self.dist.get_rank()}] recv {nbytes} B <- {peer}", flush=True)

        def f():
            t = self._view(buf, nbytes)
            self.keep.append(t)
            self.ops.append(("r", t, peer))
        return self._guard(f)

    def _end(self, ctx):
        if getattr(self, "trace", False):
            print(f"[transport {self.dist.get_rank()}] group of {len(self.ops)} ends", flush=True)

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
                for r in reqs:
                    r.wait()
            else:
                # gloo moves HOST memory: device tiles are staged through host tensors (handing gloo a device
                # pointer happens to work through the PCIe BAR mapping, and hangs now and then)
                staged = []
                reqs = []
                for kind, t, peer in self.ops:
                    h = t if self.device is None else (t.cpu() if kind == "s" else self.torch.empty(t.numel(), dtype=self.torch.uint8))
                    staged.append((kind, t, h))
                    reqs.append(self.dist.isend(h, peer, group=self.group) if kind == "s" else
                                self.dist.irecv(h, peer, group=self.group))
                for r in reqs:
                    r.wait()
                if self.device is not None:
                    for kind, t, h in staged:
                        if kind == "r":
                            t.copy_(h)
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


RCCL_ID_BYTES = 256  # include/cholmi.h: CHOL_RCCL_ID_BYTES


def install_rccl_transport(dist) -> None:
    """Build the library's own RCCL communicators (one per transport channel): rank 0 creates the id blob,
    torch.distributed (any backend; only its object broadcast is used) shares it, every rank joins."""
    from ._lib import check, lib

    L = lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = (C.c_char * RCCL_ID_BYTES)()
    box = [None]
    if rank == 0:  # a failure here must reach every rank, or they wait in the broadcast for ever
        rc = L.chol_transport_rccl_unique_id(buf)
        box = [bytes(buf) if rc == 0 else L.chol_last_error().decode(errors="replace")]
    dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], bytes):
        raise RuntimeError(f"chol_transport_rccl_unique_id failed on rank 0: {box[0]}")
    ident = (C.c_char * RCCL_ID_BYTES).from_buffer_copy(box[0])
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


def transport_selftest(self_rank: int = 0, nbytes: int = 8 << 20) -> None:
    """One message to self on each channel of the installed transport, both in flight together, byte-compared."""
    from ._lib import check, lib

    check("chol_transport_selftest", lib().chol_transport_selftest(int(self_rank), int(nbytes)))


def rehearse(N: int, B: int, P: int, Q: int, dtype: str = "f64", seed: int = 42, bump: Optional[float] = None):
    """A P x Q factorisation of plgsy(bump, seed) on ONE GPU: the ranks are threads of this process, tiles move
    by stream-ordered device copies (chol_dist_rehearse).  -> (info, descriptor holding the gathered factor, ms)"""
    from . import chameleon as ch
    from ._lib import check, lib

    cd = ch.ChamRealDouble if dtype == "f64" else ch.ChamRealFloat
    full = ch.CHAMELEON_Desc_Create(None, cd, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    ms = C.c_double()
    info = check("chol_dist_rehearse",
                 lib().chol_dist_rehearse(cd, N, B, P, Q, float(N) if bump is None else float(bump), seed, full.handle,
                                          C.byref(ms)))
    return info, full, ms.value
