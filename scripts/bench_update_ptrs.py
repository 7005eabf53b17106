#!/usr/bin/env python3
"""The task path's grouped update launch alone (chol_tile_batch, CHOL_BATCH_UPDATE): the SYRK + GEMM tasks of wave 0 of an
nt x nt tile matrix, out of place, in the client's order (rows) or the walker's (columns); beside the walker's own launch
of the same wave (chol_bench_update).  usage: bench_update_ptrs.py N B"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np, torch
from dense_linear_app_amd import chameleon as ch
from dense_linear_app_amd._lib import lib

ch.CHAMELEON_Init(1, 1)
N, B = int(sys.argv[1]), int(sys.argv[2])
nt, L = N // B, lib()
tb = B * B
d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, tb, N, N, 0, 0, N, N, 1, 1)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, d, 42)
ms, fl = ch.bench_update(d, 0, 0, 3)
print(f"walker launch wave 0 (in place, work list): {ms:.3f} ms = {fl / ms / 1e9:.2f} TFLOP/s", flush=True)
base, nbytes = d.local_ptr()
tile = lambda i, j: base + (i + j * nt) * tb * 8
out = torch.empty((nt - 1) * nt // 2 * tb, dtype=torch.float64, device="cuda")
for order in ("rows", "cols", "rows"):
    tasks = []
    if order == "rows":
        for i in range(1, nt):
            for j in range(1, i):
                tasks.append((tile(i, j), tile(i, 0), tile(j, 0)))
    else:
        for j in range(nt - 1, 0, -1):
            for i in range(j + 1, nt):
                tasks.append((tile(i, j), tile(i, 0), tile(j, 0)))
    ngemm = len(tasks)
    for i in range(1, nt):
        tasks.append((tile(i, i), tile(i, 0), 0))
    n = len(tasks)
    arr = lambda k: (C.c_void_p * n)(*[t[k] or None for t in tasks])
    outs = (C.c_void_p * n)(*[out.data_ptr() + q * tb * 8 for q in range(n)])
    cin, a, b = arr(0), arr(1), arr(2)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert L.chol_tile_batch(4, ch.ChamRealDouble, B, n, cin, a, b, outs, None, 0) == 0
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    flops = (2.0 * ngemm + (n - ngemm)) * B ** 3
    print(f"pointer-list launch, {order}: {n} tasks {best * 1e3:.3f} ms = {flops / best / 1e12:.2f} TFLOP/s (host-timed, synchronous call)", flush=True)
