#!/usr/bin/env python3
"""One rank of a p x q grid at a time on the one GPU of a test box, with a transport whose sends do nothing and whose receives
deliver zeros (chol_set_transport_null): every kernel launch, event and transport call of that rank's real schedule is issued (the numerical
result is meaningless -- the panels it would have received are missing -- but the kernels do the same work).
Reported per rank: the host time of the walker per wave (chol_dist_last_stats), the transport operations it
posted, and the DEVICE time of the rank's whole schedule (chol_last_potrf_stats): its compute with communication
taken as free -- the per-rank input of the critical-path projection in DESIGN.md section 5.
    python scripts/dist_issue_time.py [N tile]      (default 65536 1024; grids 1x1, 2x1, 2x2, 4x2 -- the defaults -- and 2x4, 8x1, every rank)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd import distributed as dd
from dense_linear_app_amd._lib import lib

N, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65536, 1024)
ch.CHAMELEON_Init(1, 1)
L = lib()
for P, Q in ((1, 1), (2, 1), (2, 2), (4, 2), (2, 4), (8, 1)):
    world = P * Q
    for rank in range(world):
        ch.set_rank(rank, world)
        d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, P, Q)
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
        L.chol_set_transport_null()
        ms = []
        for rep in range(3):
            ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
            L.chol_potrf_tile(ch.ChamLower, d.handle)  # info is meaningless here
            st = dd.dist_last_stats()
            ms.append(ch.last_potrf_stats()["total_ms"])
        print(f"grid {P}x{Q} rank {rank}: N={N} tile={B} waves={N // B}  device time of this rank's schedule {min(ms[1:]):.2f} ms; "
              f"host issue {st['issue_us_per_wave']:.1f} us/wave, "
              f"{st['sends']} sends {st['recvs']} recvs, {st['bytes_sent'] / 2**30:.2f} GiB sent by this rank", flush=True)
        ch.CHAMELEON_Desc_Destroy(d)
ch.set_rank(0, 1)
