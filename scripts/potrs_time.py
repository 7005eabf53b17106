#!/usr/bin/env python3
"""Timing of CHAMELEON_dpotrs_Tile (the solve after the factor): python scripts/potrs_time.py [N tile nrhs]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch
ch.CHAMELEON_Init(1, 1)
N, B, nrhs = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 512, 2048)
A = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, A, 42)
assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, A) == 0
X = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, nrhs, 0, 0, N, nrhs, 1, 1)
for rep in range(3):
    t0 = time.perf_counter()
    ch.CHAMELEON_dpotrs_Tile(ch.ChamLower, A, X)
    dt = time.perf_counter() - t0
    print(f"potrs N={N} tile={B} nrhs={nrhs}: {dt * 1e3:.1f} ms  {2.0 * N * N * nrhs / dt / 1e12:.1f} TFLOP/s", flush=True)
