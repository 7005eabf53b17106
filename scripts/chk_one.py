#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import chameleon as ch
ch.CHAMELEON_Init(1, 1)
N, NB = int(sys.argv[1]), int(sys.argv[2])
out = []
for rep in range(4):
    d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, NB, NB, NB * NB, N, N, 0, 0, N, N, 1, 1)
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
    out.append(f"{ch.residual_plgsy(d, float(N), 42):.1e}")
    ch.CHAMELEON_Desc_Destroy(d)
print(os.environ.get("TAG", ""), N, NB, out, flush=True)
