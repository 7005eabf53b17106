#!/usr/bin/env python3
"""Larger one-GPU rehearsals of the p x q walker (chol_dist_rehearse): many waves, every grid, both precisions;
residual of the gathered factor.   python scripts/rehearse_stress.py [N tile]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd import distributed as dd

ch.CHAMELEON_Init(1, 1)
N, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 512)
for dtype, tol in (("f64", 1e-13), ("f32", 5e-5)):
    for P, Q in ((1, 1), (2, 1), (2, 2), (4, 2), (2, 4), (8, 1), (3, 2)):
        info, full, ms = dd.rehearse(N, B, P, Q, dtype)
        res = ch.residual_plgsy(full, float(N), 42)
        ch.CHAMELEON_Desc_Destroy(full)
        st = dd.dist_last_stats()
        print(f"{dtype} grid {P}x{Q} N={N} tile={B}: info={info} residual={res:.2e} ({'ok' if info == 0 and res <= tol else 'FAIL'}) "
              f"wall {ms:.1f} ms ({N**3 / 3 / ms / 1e9:.1f} TFLOP/s with all ranks on one GPU), rank 0: {st['sends']} sends {st['recvs']} recvs", flush=True)
