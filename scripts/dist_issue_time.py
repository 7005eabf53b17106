#!/usr/bin/env python3
"""Host-side issue time of the C++ distributed wave loop (csrc/dist.hip), per wave.
RCCL needs one GPU per rank, so on a one-GPU box this process plays ONE rank of a p x q grid with a
transport whose sends / receives do nothing: every kernel launch, event and transport call of that
rank's real schedule is issued (the numerical result is meaningless -- the panels it would have
received are missing), and the host time of the loop is what chol_dist_last_stats reports.
    python scripts/dist_issue_time.py [N tile]      (default 65536 1024, grids 1x2, 2x2, 2x4)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd import distributed as dd
from dense_linear_app_amd._lib import lib

N, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65536, 1024)
T = dd._TransportTable
f = dict(T._fields_)
noop = T(None, f["group_begin"](lambda c: 0), f["send"](lambda c, b, n, p, s: 0), f["recv"](lambda c, b, n, p, s: 0),
         f["group_end"](lambda c: 0), f["allreduce_max"](lambda c, v: 0))
ch.CHAMELEON_Init(1, 1)
L = lib()
for world in (2, 4, 8):
    P, Q = dd.grid_for(world)
    for rank in (0, world - 1):
        ch.set_rank(rank, world)
        d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, P, Q)
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
        L.chol_set_transport(C.byref(noop))
        for rep in range(2):
            L.chol_potrf_tile(ch.ChamLower, d.handle)  # info is meaningless here
            st = dd.dist_last_stats()
        print(f"grid {P}x{Q} rank {rank}: N={N} tile={B} waves={N // B}  host issue {st['issue_us_per_wave']:.1f} us/wave, "
              f"{st['sends']} sends {st['recvs']} recvs, {st['bytes_sent'] / 2**30:.2f} GiB sent by this rank", flush=True)
        ch.CHAMELEON_Desc_Destroy(d)
ch.set_rank(0, 1)
