#!/usr/bin/env python3
"""How long does a diagonal-block workgroup itself run under a concurrent trailing update?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd._lib import lib

ch.CHAMELEON_Init(1, 1)
L = lib()
N, B = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16384x1024").split("x"))
d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
L.chol_debug_stamps(1, None, 0)
ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
buf = (C.c_ulonglong * 8000)()
n = L.chol_debug_stamps(0, buf, 1000)
a = np.array(buf[:8 * n], dtype=np.uint64).reshape(n, 8).astype(np.int64)
a = a[np.argsort(a[:, 0])]
us = lambda x: x / 100.0
tot = us(a[:, 1] - a[:, 0]); load = us(a[:, 2] - a[:, 0]); pa = us(a[:, 3]); pb = us(a[:, 4])
lst = us(a[:, 5] - a[:, 2]) - pa - pb; wd = us(a[:, 6] - a[:, 5]); pc = us(a[:, 7] - a[:, 6]); st = us(a[:, 1] - a[:, 7])
alone = tot < 1.2 * tot.min()
for name, m in (("alone", alone), ("contended (>2.5x)", tot > 2.5 * tot.min())):
    if m.sum() == 0: continue
    f = lambda v: f"{np.median(v[m]):7.1f}"
    print(f"{name:18s} n={m.sum():3d} total={f(tot)} load={f(load)} phaseA={f(pa)} phaseB={f(pb)} storeL={f(lst)} Wd={f(wd)} phaseC={f(pc)} storeW={f(st)}  (median us)")
