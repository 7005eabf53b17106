#!/usr/bin/env python3
"""Per-wave critical-path model of the p x q factorisation: which grid for 8 MI355X?
Inputs (measured on one GPU unless said otherwise; override on the command line key=value):
  t_tile   us per tile update (2 B^3 flops) inside the DAG            [B=1024 fp64: 31.6 at 68 TFLOP/s]
  t_trsm   us per panel tile of the TRSM                              [43: B^3 flops at ~25 TFLOP/s]
  t_potrf  us per diagonal tile (B/128 chain steps of ~72 us)         [580]
  t_syrk   us for the SYRK on the next diagonal tile                  [30]
  t_hop    us per cross-stream / transport hand-over on the chain     [15]
  bw       GB/s one xGMI link carries in one direction (ASSUMED: never measured here)   [60]
  lat      us per transport group (launch + handshake; ASSUMED)       [25]
Per wave k (m = nt-1-k tiles below the diagonal tile), for every rank: its tiles of the update; the cycle
  A = head TRSM -> (head tile to the next diagonal owner, unless it is the same rank: q = 1) -> SYRK -> POTRF -> L(k,k) to its column
  B = local TRSM of the panel part (m/p tiles) -> exchange (critical link: the part to one row peer, or with q = 1 to every
      other rank, m/p tiles either way) -> column k+1 (m/p tiles per rank of its process column)
wave time = max(max over ranks of the update, A, B); the walker overlaps A, B of wave k+1 with the update of wave k."""
import sys

par = dict(nt=64, B=1024, esize=8, t_tile=31.6, t_trsm=43.0, t_potrf=580.0, t_syrk=30.0, t_hop=15.0, bw=60.0, lat=25.0, t1=1357.0)
for a in sys.argv[1:]:
    k, v = a.split("=")
    par[k] = float(v)
nt = int(par["nt"])
tile_us = par["B"] ** 2 * par["esize"] / (par["bw"] * 1e9) * 1e6  # one tile over one link


def run(P, Q):
    tot = upd_only = 0.0
    chain_bound = 0
    for k in range(nt - 1):
        m = nt - 1 - k
        # tiles of wave k's update per rank (i > k, k < j <= i), diagonal tiles count half
        load = [[0.0] * Q for _ in range(P)]
        for j in range(k + 1, nt):
            for i in range(j, nt):
                load[i % P][j % Q] += 0.5 if i == j else 1.0
        upd = max(max(r) for r in load) * par["t_tile"]
        part = -(-m // P)  # tiles of the panel a process row holds
        head_send = 0.0 if Q == 1 else tile_us + par["lat"] + par["t_hop"]
        lkk_send = 0.0 if P == 1 else 1.125 * tile_us + par["lat"] + par["t_hop"]
        A = par["t_trsm"] + head_send + par["t_syrk"] + par["t_hop"] + par["t_potrf"] + lkk_send
        exch = 0.0 if P * Q == 1 else part * tile_us + par["lat"]
        B = part * par["t_trsm"] + exch + part * par["t_tile"] + 2 * par["t_hop"]
        w = max(upd, A, B)
        chain_bound += w > upd
        tot += w
        upd_only += upd
    tot += par["t_potrf"]
    return tot / 1e3, upd_only / 1e3, chain_bound


print(f"nt={nt} tile {par['B']:.0f}: one tile over one link {tile_us:.0f} us at {par['bw']:.0f} GB/s; 1 GPU measured {par['t1']:.0f} ms")
for P, Q in ((1, 1), (1, 2), (2, 1), (2, 2), (4, 1), (2, 4), (4, 2), (8, 1), (1, 8)):
    ms, upd, cb = run(P, Q)
    print(f"grid {P}x{Q}: projected {ms:7.1f} ms  ({par['t1'] / ms:4.2f}x of the measured 1-GPU time; updates alone {upd:6.1f} ms, "
          f"{cb} of {nt - 1} waves bound by the panel cycle)")
