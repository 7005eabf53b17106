#!/usr/bin/env python3
"""Race screen for the flow form of the tile POTRF: chain-bound shapes (its default rule applies), many repetitions, residual and the number
of flow-form waves checked every time."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd._lib import lib
ch.CHAMELEON_Init(1, 1)
t0 = time.time()
for N, B, dt, reps in [(2048, 512, "f64", 300), (3072, 384, "f64", 200), (4096, 512, "f64", 200), (6144, 512, "f64", 100), (5120, 512, "f32", 100),
                       (1536, 512, "f64", 300), (3000, 448, "f64", 100), (4608, 512, "f64", 150)]:
    dtype = ch.ChamRealDouble if dt == "f64" else ch.ChamRealFloat
    tol = 1e-13 if dt == "f64" else 5e-5
    d = ch.CHAMELEON_Desc_Create(None, dtype, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    w, fw = 0.0, set()
    for r in range(reps):
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 2000 + r)
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        fw.add(lib().chol_debug_flow_waves())
        res = ch.residual_plgsy(d, float(N), 2000 + r)
        w = max(w, res)
        if info != 0 or not (res <= tol):
            print(f"FAIL N={N} B={B} {dt} rep={r} info={info} residual={res}", flush=True); sys.exit(1)
    print(f"ok N={N} B={B} {dt} reps={reps} flow waves per run={sorted(fw)} worst residual={w:.2e} ({time.time()-t0:.0f} s)", flush=True)
print("stress ok")
