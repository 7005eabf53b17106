#!/usr/bin/env python3
"""Measured accuracy of the GPU tile operations against the CPU oracle as a function of the
condition number (the table of DESIGN.md section 6).  Run on the GPU box:
    python scripts/cond_table.py > gpurun_out/cond_table.txt
Columns: componentwise backward error of POTRF and TRSM in units of B*eps (GPU | oracle), i.e.
max |L L^T - A| / (B eps |L||L^T|) and max |X L^T - A| / (B eps (|X||L^T| + |A|)); forward difference
GPU vs oracle relative to max|ref|, and that difference divided by (B eps kappa)."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dense_linear_app_amd import chameleon as ch  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker)
import test_gpu_conditioning as tc  # noqa: E402

EPS = np.finfo(float).eps


def bw_potrf(L, A, B):
    L = np.tril(L)
    return float((np.abs(np.tril(L @ L.T - A)) / (B * EPS * (np.abs(L) @ np.abs(L).T) + 1e-300)).max())


def bw_trsm(X, L, A, B):
    L = np.tril(L)
    return float((np.abs(X @ L.T - A) / (B * EPS * (np.abs(X) @ np.abs(L).T + np.abs(A)) + 1e-300)).max())


def main():
    ch.CHAMELEON_Init(1, 1)
    print("kind       B    kappa2(A)  kinf(L)   potrf_bw gpu|orc   trsm_bw gpu|orc   |dL|/|L|   /(B eps k)  |dX|/|X|   /(B eps kL)")
    cases = [("spectral", B, k) for B in (128, 512, 1024) for k in (1e2, 1e6, 1e10)] + [("spectral", 128, 1e13)]
    cases += [("graded", B, k) for B in (128, 512) for k in (13, 26)]
    for kind, B, k in cases:
        if kind == "spectral":
            A = tc.spd_spectral(2 * B, k, seed=B + int(np.log10(k)))
        else:
            A, _, _ = tc.spd_graded(2 * B, k, seed=B + int(k))
        Akk, A10 = np.asfortranarray(A[:B, :B]), np.asfortranarray(A[B:, :B])
        L = Akk.copy(order="F")
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, tc.desc1(ch, L))
        Lref, iref = orc.dpotrf(Akk)
        X = A10.copy(order="F")
        ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, tc.desc1(ch, L), tc.desc1(ch, X))
        Lt = np.asfortranarray(np.tril(L))
        Xref = orc.dtrsm(Lt, A10)
        kA, kL = np.linalg.cond(Akk), np.linalg.cond(Lt, np.inf)
        dL = np.abs(np.tril(L) - np.tril(Lref)).max() / np.abs(Lref).max()
        dX = np.abs(X - Xref).max() / np.abs(Xref).max()
        print(f"{kind:9s} {B:5d} {kA:10.2e} {kL:9.2e}  {bw_potrf(L, Akk, B):7.3f} | {bw_potrf(Lref, Akk, B):5.3f}   "
              f"{bw_trsm(X, Lt, A10, B):7.3f} | {bw_trsm(Xref, Lt, A10, B):5.3f}  {dL:9.2e} {dL / (B * EPS * kA):9.2e}  {dX:9.2e} {dX / (B * EPS * kL):9.2e}"
              f"  info {info}/{iref}", flush=True)
    # general triangular factor (TRSM alone)
    for kl in (1e3, 1e8, 1e12):
        B = 256
        rng = np.random.default_rng(int(np.log10(kl)))
        U, _ = np.linalg.qr(rng.standard_normal((B, B)))
        V, _ = np.linalg.qr(rng.standard_normal((B, B)))
        _, R = np.linalg.qr(((U * np.logspace(0, -np.log10(kl), B)) @ V.T).T)
        Lt = np.asfortranarray(np.tril(R.T))
        A = np.asfortranarray(rng.uniform(-1, 1, (B, B)))
        X = A.copy(order="F")
        ch.CHAMELEON_dtrsm_Tile(ch.ChamRight, ch.ChamLower, ch.ChamTrans, ch.ChamNonUnit, 1.0, tc.desc1(ch, Lt.copy(order="F")), tc.desc1(ch, X))
        Xref = orc.dtrsm(Lt, A)
        kL = np.linalg.cond(Lt, np.inf)
        dX = np.abs(X - Xref).max() / np.abs(Xref).max()
        print(f"triangular {B:4d} {'-':>10s} {kL:9.2e}  {'-':>7s} | {'-':>5s}   {bw_trsm(X, Lt, A, B):7.3f} | {bw_trsm(Xref, Lt, A, B):5.3f}  "
              f"{'-':>9s} {'-':>9s}  {dX:9.2e} {dX / (B * EPS * kL):9.2e}", flush=True)
    # whole matrix
    for N, B, k in ((2048, 512, 1e2), (2048, 512, 1e8), (2048, 512, 1e11), (4096, 1024, 1e10)):
        A = tc.spd_spectral(N, k, seed=N + int(np.log10(k)))
        d = ch.CHAMELEON_Desc_Create(None, ch.ChamRealDouble, B, B, B * B, N, N, 0, 0, N, N, 1, 1)
        d.from_lapack(A)
        info = ch.CHAMELEON_dpotrf_Tile(ch.ChamLower, d)
        L = np.tril(d.to_lapack())
        Lref, iref = orc.cholesky_lower(A, B)
        nA = np.linalg.norm(A)
        print(f"whole N={N} B={B} kappa={k:.0e}: residual_F gpu {np.linalg.norm(L @ L.T - A) / nA:.2e} | oracle {np.linalg.norm(Lref @ Lref.T - A) / nA:.2e}; "
              f"componentwise/(N eps) gpu {bw_potrf(L, A, N):.3f} | oracle {bw_potrf(Lref, A, N):.3f}; max|dL|/max|L| {np.abs(L - Lref).max() / np.abs(Lref).max():.2e}; info {info}/{iref}", flush=True)


if __name__ == "__main__":
    main()
