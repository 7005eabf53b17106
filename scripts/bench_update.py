#!/usr/bin/env python3
"""Kernel-level micro-benchmark: the trailing-update launch of wave 0 alone, with ablations."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch

ch.CHAMELEON_Init(1, 1)
cfgs = [tuple(map(int, a.split("x"))) for a in sys.argv[1:]] or [(32768, 1024)]
for N, B in cfgs:
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    D = 0x100  # diagnostic kernel (register staging) with ablation bits
    variants = [(0, "production (paired, LDS-DMA)"), (D, "diag baseline"), (0, "production again")]
    if os.environ.get("ABLATE") == "1":
        variants += [(D | 32, "diag: cache-hot global loads"), (D | 4, "diag: no C read"), (D | 1, "diag: no global loads"), (D | 2, "diag: no LDS reads"),
                     (D | 8, "diag: no barrier"), (D | 3, "diag: no gl+lds"), (D | 15, "diag: mfma+store only")]
    for abl, name in variants:
        try:
            ms, tf = ch.bench_update(d, 0, abl, 3)
        except ch.CholmiError as e:  # ablation twin: diagnostic build only (make -C dense_linear_app_amd/csrc DIAG=1)
            print(f"N={N} B={B} ablate={abl:2d} {name:28s} skipped: {e.msg}", flush=True)
            continue
        print(f"N={N} B={B} ablate={abl:2d} {name:28s} {ms:8.3f} ms  {tf:6.2f} TF/s", flush=True)
    ch.CHAMELEON_Desc_Destroy(d)
