#!/usr/bin/env python3
"""Quick single-GPU timing probe of the resident whole-matrix POTRF (not the bench)."""
import sys
import time

import os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch


def run(N, B, reps=3, dtype=None, check=True):
    dtype = dtype or ch.ChamRealDouble
    d = ch.CHAMELEON_Desc_Create(None, dtype, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    best = 1e9
    for r in range(reps + 1):
        ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
        t0 = time.perf_counter()
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        dt = time.perf_counter() - t0
        st = ch.last_potrf_stats()
        try:
            from dense_linear_app_amd import distributed as _dd
            st["issue"] = _dd.dist_last_stats()["issue_us_per_wave"]
        except Exception:
            st["issue"] = -1.0
        if r > 0:
            best = min(best, dt)
        print(f"N={N} B={B} {'f32' if dtype == ch.ChamRealFloat else 'f64'} rep={r} info={info} wall={dt*1e3:.2f} ms dev={st['total_ms']:.2f} ms "
              f"{N**3/3/dt/1e12:.2f} TF/s  upd_ms={st['update_ms']:.2f} upd_tf={st['update_flops']/max(st['update_ms'],1e-9)/1e9:.2f} host_issue_us_per_wave={st['issue']:.0f}", flush=True)
    if check:
        print("  residual", ch.residual_plgsy(d, float(N), 42), flush=True)
    ch.CHAMELEON_Desc_Destroy(d)
    return best


if __name__ == "__main__":
    ch.CHAMELEON_Init(1, 1)
    for w in ((1, 2) if os.environ.get("PROBE_QUICK") != "1" else ()):
        print(f"mfma probe f64 waves/simd={w}: {ch.mfma_probe(ch.ChamRealDouble, w):.2f} TF/s ; f32: {ch.mfma_probe(ch.ChamRealFloat, w):.2f} TF/s", flush=True)
    cfgs = [(4096, 512), (16384, 512), (16384, 1024), (32768, 1024)]
    if len(sys.argv) > 1:
        cfgs = [tuple(a.split("x")) for a in sys.argv[1:]]
    quick = os.environ.get("PROBE_QUICK") == "1"
    for cfg in cfgs:
        N, B = int(cfg[0]), int(cfg[1])
        dt = ch.ChamRealFloat if len(cfg) > 2 and cfg[2] == "f32" else ch.ChamRealDouble
        if quick:
            ch.set_profiling(False)
            run(N, B, reps=0, dtype=dt, check=False)
            continue
        ch.set_profiling(False)
        run(N, B, reps=2, dtype=dt, check=(N <= 16384))
        ch.set_profiling(True)
        run(N, B, reps=1, dtype=dt, check=False)
