#!/usr/bin/env python3
"""In-kernel stamps of the flow form of the tile POTRF (k_flow_factor): per wave of the factorisation, when each
diagonal-block workgroup entered, started to factor (its block updated by every earlier step), and ended; the step of
the chain is the distance between two consecutive factor starts.  usage: flow_stamps.py NxB [waves to print]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd._lib import lib

ch.CHAMELEON_Init(1, 1)
L = lib()
N, B = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8192x512").split("x"))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nbm = B // 128
d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
L.chol_debug_stamps(1, None, 0)
info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
buf = (C.c_ulonglong * 8000)()
n = L.chol_debug_stamps(0, buf, 1000)
a = np.array(buf[:8 * n], dtype=np.uint64).reshape(n, 8).astype(np.int64)
us = lambda x: x / 100.0
# columns: t0, end, loaded (= factor start), tA, tB, L stored, Wd done, inverse done
a = a[(a[:, 2] <= 0) | (a[:, 2] >= 100)]  # (records of the step kernels carry a small tag there: scripts/flow_marks.py)
n = len(a)
a = a[np.argsort(a[:, 2])]  # by factor start: the order of the chain
t_base = a[0, 0]
print(f"# N={N} tile={B} info={info}: {n} diagonal-block workgroups stamped; times in us from the first entry")
print("#  idx    entry  factor_start     end   | step (start-to-start)  phaseA  phaseB  wait_before_factor  tail(after B)")
steps = []
for i in range(n):
    t0, end, fs, tA, tB = a[i, 0], a[i, 1], a[i, 2], a[i, 3], a[i, 4]
    step = us(a[i, 2] - a[i - 1, 2]) if i else 0.0
    steps.append(step)
    if i < nshow * nbm or i >= n - nshow * nbm:
        print(f"  {i:4d} {us(t0 - t_base):9.1f} {us(fs - t_base):9.1f} {us(end - t_base):9.1f}  | {step:8.1f} {us(tA):8.1f} {us(tB):7.1f} {us(fs - t0):9.1f} {us(end - fs) - us(tA) - us(tB):9.1f}")
steps = np.array(steps[1:])
intile = np.array([s for i, s in enumerate(steps, 1) if i % nbm != 0])
cross = np.array([s for i, s in enumerate(steps, 1) if i % nbm == 0])
if len(intile):
    print(f"# in-tile steps: median {np.median(intile):.1f} us (min {intile.min():.1f}, p90 {np.percentile(intile, 90):.1f}); "
          f"tile-crossing steps: median {np.median(cross):.1f} us" if len(cross) else "")
print(f"# phase A median {np.median(us(a[:, 3])):.1f} us, phase B median {np.median(us(a[:, 4])):.1f} us")
