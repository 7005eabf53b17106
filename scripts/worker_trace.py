#!/usr/bin/env python3
"""One run of the task API (client -> control plane -> worker -> executor) for a kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/worker_trace.py N B
then  python3 scripts/worker_trace.py --summary <kernel_trace.csv>  (the run is what follows the last idle gap > 0.2 s)."""
import csv, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


if sys.argv[1] == "--summary":
    rows = []
    for r in csv.DictReader(open(sys.argv[2])):
        n = r["Kernel_Name"].split("(")[0].replace("void cholmi::", "").replace("cholmi::", "").strip()
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(r[1] for r in rows[max(0, i - 50):i]) > 200_000_000:
            cut = i
    rows = rows[cut:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print(f"span {1e-6 * (t1 - t0):.3f} ms, {len(rows)} kernels, busy (union of all) {1e-6 * union([(s, e) for s, e, _ in rows]):.3f} ms")
    names = {}
    for s, e, n in rows:
        names.setdefault(n, []).append((s, e))
    for n, iv in sorted(names.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        tot = sum(e - s for s, e in iv)
        print(f"  {n[:60]:60s} n={len(iv):5d} sum {1e-6 * tot:9.3f} ms  union {1e-6 * union(iv):9.3f} ms  avg {1e-3 * tot / len(iv):8.1f} us")
    upd = [(s, e) for s, e, n in rows if n.startswith("k_update_ptrs") or n.startswith("k_trail_update")]
    chain = [(s, e) for s, e, n in rows if not (n.startswith("k_update_ptrs") or n.startswith("k_trail_update"))]
    print(f"update kernels: union {1e-6 * union(upd):.3f} ms; everything else: union {1e-6 * union(chain):.3f} ms; "
          f"time with NO update kernel running {1e-6 * ((t1 - t0) - union(upd)):.3f} ms")
    # the longest launches of the update kernel, in order: start offset, duration
    if len(sys.argv) > 3:
        for s, e, n in rows:
            print(f"{1e-3 * (s - t0):10.1f} {1e-3 * (e - s):9.1f} {n[:50]}")
    raise SystemExit(0)

import numpy as np
import bench
from dense_linear_app_amd import chameleon as ch

ch.CHAMELEON_Init(1, 1)
N, B = int(sys.argv[1]), int(sys.argv[2])
r = bench.worker_path(N, B)
time.sleep(0.5)
from dense_linear_app_amd import client
rng = np.random.default_rng(7)
A = np.zeros((N, N), order="F")
for j in range(0, N, B):
    A[j:, j:j + B] = rng.uniform(-0.5, 0.5, size=(N - j, min(B, N - j)))
A[np.diag_indices(N)] += float(N)
import torch
torch.cuda.synchronize()
time.sleep(0.5)
res = client.run_cholesky_dag(N, B, A=A, device_results=True, batched=True)
print(f"traced run: {res.seconds * 1e3:.2f} ms = {N ** 3 / 3 / res.seconds / 1e12:.2f} TFLOP/s (untraced best of 3: {r['tflops']})")
