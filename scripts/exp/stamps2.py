import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd._lib import lib
ch.CHAMELEON_Init(1, 1)
L = lib()
N, B = 16384, 1024
d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
for r in range(2):
    ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
    if r == 1: L.chol_debug_stamps(1, None, 0)
    ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
buf = (C.c_ulonglong * 8000)()
n = L.chol_debug_stamps(0, buf, 1000)
a = np.array(buf[:8 * n], dtype=np.uint64).reshape(n, 8)
tot = (a[:, 1] - a[:, 0]).astype(np.int64) / 100.0
alone = tot < 1.2 * tot.min()
tA = (a[:, 3] & 0xffffffff).astype(np.int64) / 100.0
w3 = (a[:, 3] >> 32).astype(np.int64) / 100.0
w0 = (a[:, 4] & 0xffffffff).astype(np.int64) / 100.0
w1 = (a[:, 4] >> 32).astype(np.int64) / 100.0
m = alone
inv_tail = (a[:, 6] & 0xffffffff).astype(np.int64) / 100.0
row_tail = (a[:, 6] >> 32).astype(np.int64) / 100.0
print(f"tails after the last flag (sum of 8 panels): inverse {np.median(inv_tail[m]):.2f} rows {np.median(row_tail[m]):.2f} us")
print(f"alone n={m.sum()} total {np.median(tot[m]):.1f} phaseA {np.median(tA[m]):.1f}  wave0 loop {np.median(w0[m]):.1f}  wave1 follow {np.median(w1[m]):.1f}  wave3 follow {np.median(w3[m]):.1f} us")
