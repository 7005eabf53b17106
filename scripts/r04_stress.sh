#!/bin/bash
# round 4: race screen with the flow form where its default rule applies, two processes side by side, and the small chain-bound shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY' > gpurun_out/r04_stress_small.txt 2>&1 &
import os, sys, time
sys.path.insert(0, ".")
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd._lib import lib
ch.CHAMELEON_Init(1, 1)
t0 = time.time()
for N, B, dt, reps in [(2048, 512, "f64", 150), (3072, 384, "f64", 100), (4096, 512, "f64", 100), (6144, 512, "f64", 60), (5120, 512, "f32", 60), (1536, 512, "f64", 150), (3000, 448, "f64", 60)]:
    dtype = ch.ChamRealDouble if dt == "f64" else ch.ChamRealFloat
    tol = 1e-13 if dt == "f64" else 5e-5
    d = ch.CHAMELEON_Desc_Create(None, dtype, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    w, fw = 0.0, 0
    for r in range(reps):
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 2000 + r)
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        fw = lib().chol_debug_flow_waves()
        res = ch.residual_plgsy(d, float(N), 2000 + r)
        w = max(w, res)
        if info != 0 or not (res <= tol):
            print(f"FAIL N={N} B={B} {dt} rep={r} info={info} residual={res}", flush=True); sys.exit(1)
    print(f"ok N={N} B={B} {dt} reps={reps} flow waves per run={fw} worst residual={w:.2e} ({time.time()-t0:.0f} s)", flush=True)
print("stress ok")
PY
p1=$!
python scripts/stress.py > gpurun_out/r04_stress_std.txt 2>&1
echo "std rc=$?"; wait $p1; echo "small rc=$?"
tail -9 gpurun_out/r04_stress_small.txt; tail -9 gpurun_out/r04_stress_std.txt
examples/v6_driver 1 1 4096 512 512 512 262144 4096 4096 0 0 4096 4096 1 1 42 2>&1 | tail -3
