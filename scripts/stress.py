#!/usr/bin/env python3
"""Race screen: many factorisations at several shapes, residual checked every time."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch

ch.CHAMELEON_Init(1, 1)
cfgs = [(4096, 512, "f64", 120), (8192, 1024, "f64", 60), (6144, 256, "f64", 40), (5000, 448, "f64", 30),
        (8192, 512, "f32", 40), (16384, 1024, "f64", 25), (12288, 128, "f64", 10)]
if os.environ.get("STRESS_BIG") == "1":  # the compute-bound regime: panel kernels beside a running update
    cfgs += [(32768, 1024, "f64", 12), (32768, 512, "f64", 8), (65536, 1024, "f64", 3), (65536, 1024, "f32", 4)]
worst = {}
t0 = time.time()
for N, B, dt, reps in cfgs:
    dtype = ch.ChamRealDouble if dt == "f64" else ch.ChamRealFloat
    tol = 1e-13 if dt == "f64" else 5e-5
    d = ch.CHAMELEON_Desc_Create(None, dtype, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    w = 0.0
    for r in range(reps):
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 1000 + r)
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        res = ch.residual_plgsy(d, float(N), 1000 + r)
        w = max(w, res)
        if info != 0 or not (res <= tol):
            print(f"FAIL N={N} B={B} {dt} rep={r} info={info} residual={res}", flush=True)
            sys.exit(1)
    print(f"ok N={N} B={B} {dt} reps={reps} worst residual={w:.2e}  ({time.time()-t0:.0f} s)", flush=True)
    ch.CHAMELEON_Desc_Destroy(d)
print("stress ok")
