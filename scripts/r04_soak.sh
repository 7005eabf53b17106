#!/bin/bash
# round 4: a longer race screen of the final build (chain-bound shapes three times over, the standard stress twice, mid sizes)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do timeout -k 10 300 python scripts/stress_flow.py 2>&1 | tail -1; done
for i in 1 2; do timeout -k 10 400 python scripts/stress.py 2>&1 | tail -1; done
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
from dense_linear_app_amd import chameleon as ch
ch.CHAMELEON_Init(1, 1)
t0 = time.time()
for N, B, reps in [(8192, 512, 150), (7168, 512, 150), (9216, 512, 100), (10240, 512, 80), (12288, 512, 50), (6144, 384, 150), (8192, 384, 100), (16384, 512, 25), (8192, 1024, 60)]:
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    w = 0.0
    for r in range(reps):
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 3000 + r)
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        res = ch.residual_plgsy(d, float(N), 3000 + r)
        w = max(w, res)
        if info != 0 or not (res <= 1e-13):
            print(f"FAIL N={N} B={B} rep={r} info={info} residual={res}", flush=True); sys.exit(1)
    ch.CHAMELEON_Desc_Destroy(d)
    print(f"ok N={N} B={B} reps={reps} worst residual={w:.2e} ({time.time()-t0:.0f} s)", flush=True)
print("soak ok")
PY
