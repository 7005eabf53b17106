#!/usr/bin/env python3
"""Trailing-update launch of wave k alone, for several k (CHOLMI_LIST_ORDER selects the tile order)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch
ch.CHAMELEON_Init(1, 1)
N, B = int(sys.argv[1]), int(sys.argv[2])
d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
tot = 0.0
for k in [int(a) for a in sys.argv[3:]]:
    ms, tf = ch.bench_update(d, k, 0, 3)
    tot += ms
    print(f"{os.environ.get('CHOLMI_LIST_ORDER','split'):12s} N={N} B={B} k={k:3d} {ms:8.3f} ms {tf:6.2f} TF/s", flush=True)
print("sum ms", round(tot, 3))
