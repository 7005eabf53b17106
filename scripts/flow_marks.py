#!/usr/bin/env python3
"""Device-side timeline of the flow form's kernels from in-kernel stamps (no profiler): usage flow_marks.py NxB t0_us t1_us"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd._lib import lib

ch.CHAMELEON_Init(1, 1)
L = lib()
N, B = (int(x) for x in sys.argv[1].split("x"))
lo, hi = float(sys.argv[2]), float(sys.argv[3])
d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
for _ in range(2):
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
L.chol_debug_stamps(1, None, 0)
info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
buf = (C.c_ulonglong * 8000)()
n = L.chol_debug_stamps(0, buf, 1000)
a = np.array(buf[:8 * n], dtype=np.uint64).reshape(n, 8).astype(np.int64)
names = {1: "rows_i", 2: "rows_p", 3: "E_i", 4: "E_p", 5: "slice_d", 6: "slice_o"}
ev = []
for r in a:
    if 0 < r[2] < 100:
        t = int(r[2])
        nm = names.get(t) or (f"rows step {t - 10} block row" if t < 20 else f"head step {t - 20} block row")
        ev.append((r[0], r[1], f"{nm}({int(r[3])})"))
    else:  # a diagonal-block workgroup: entry, end, factor start
        ev.append((r[0], r[1], f"F  factor_start=+{(r[2] - r[0]) / 100:.1f} A={r[3] / 100:.1f} B={r[4] / 100:.1f}"))
ev.sort()
t0 = ev[0][0]
print(f"# N={N} tile={B} info={info} {n} records")
for s, e, nm in ev:
    if lo <= (s - t0) / 100 <= hi:
        print(f"{(s - t0) / 100:9.1f} {(e - t0) / 100:9.1f} {(e - s) / 100:7.1f}  {nm}")
