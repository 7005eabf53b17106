#!/usr/bin/env python3
"""BASELINE configs 4 and 5 with their 8-GPU partitioning at FULL size on ONE GPU: the p x q walker of every rank (threads of
this process, stream-ordered copies for the transport, no device synchronisation: chol_dist_rehearse), the gathered factor
checked by its residual and against the one-GPU walker's factor of the same matrix.  usage: rehearse_full_size.py [f64|f32]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
import torch
from dense_linear_app_amd import chameleon as ch, distributed as dd

ch.CHAMELEON_Init(1, 1)
which = sys.argv[1] if len(sys.argv) > 1 else "f64"
N, B = (65536, 1024) if which == "f64" else (131072, 1024)
dt = ch.ChamRealDouble if which == "f64" else ch.ChamRealFloat
tol_res, tol_l = (1e-13, 1e-12) if which == "f64" else (5e-5, 1e-4)
ref = ch.CHAMELEON_Desc_Create(None, dt, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
ch.CHAMELEON_dplgsy_Tile(float(N), ch.ChamLower, ref, 42)
t0 = time.perf_counter(); assert ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, ref) == 0; t1 = time.perf_counter() - t0
print(f"{which} N={N} tile={B}: one-GPU walker {t1 * 1e3:.0f} ms", flush=True)
if which != "f64":  # (fp32, N=131072: the one-GPU factor is not compared -- 64 GiB per matrix -- and makes room)
    ch.CHAMELEON_Desc_Destroy(ref)
    ref = None
for (P, Q) in ((4, 2), (2, 4)):
    t0 = time.perf_counter()
    info, full, ms = dd.rehearse(N, B, P, Q, which)
    wall = time.perf_counter() - t0
    res = ch.residual_plgsy(full, float(N), 42)
    # the gathered factor against the one-GPU factor on the device: max|tril(L) - tril(Lref)| / max|tril(Lref)| by the
    # driver's own operations (dlacpy Lower into zeroed matrices, dgeadd, dlange Max); fp64 only (four matrices in HBM)
    worst = float("nan")
    if which == "f64":
        t1 = ch.CHAMELEON_Desc_Create(None, dt, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
        t2 = ch.CHAMELEON_Desc_Create(None, dt, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
        for t in (t1, t2):
            ch.CHAMELEON_dgeadd_Tile(ch.ChamNoTrans, 0.0, t, 0.0, t)  # (0 * t: zero; fresh HBM of this process holds zeros or finite values)
        ch.CHAMELEON_dlacpy_Tile(ch.ChamLower, ref, t1)
        ch.CHAMELEON_dlacpy_Tile(ch.ChamLower, full, t2)
        scale = ch.CHAMELEON_dlange_Tile(ch.ChamMaxNorm, t1)
        ch.CHAMELEON_dgeadd_Tile(ch.ChamNoTrans, -1.0, t2, 1.0, t1)
        worst = ch.CHAMELEON_dlange_Tile(ch.ChamMaxNorm, t1) / scale
        ch.CHAMELEON_Desc_Destroy(t1)
        ch.CHAMELEON_Desc_Destroy(t2)
    print(f"  {P}x{Q} rehearsal: info={info} residual={res:.2e} (<= {tol_res})  max|dL|/max|L| against the one-GPU factor = {worst:.2e} (<= {tol_l})  "
          f"[{wall:.1f} s wall incl. generation and gather; device {ms:.0f} ms]", flush=True)
    assert info == 0 and res <= tol_res and not (worst > tol_l)
    ch.CHAMELEON_Desc_Destroy(full)
    torch.cuda.empty_cache()
