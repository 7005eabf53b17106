#!/usr/bin/env python3
"""Re-run the reference's VM sweep grid (benchmark.c:76-103: N x NB, run 0 = warm-up, runs 1-7
measured) on one MI355X and print the median GFLOP/s per (N, NB) and the best NB per N."""
import io, os, statistics, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from dense_linear_app_amd import driver

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/reference_grid.csv"
if os.path.exists(out):
    os.remove(out)
Ns = [1000, 5000, 8000, 12000, 16000]
NBs = list(range(128, 513, 64))
rows = driver.bench(Ns, NBs, csv_path=out, repeats=8, out=io.StringIO())
import json
gold = {(c["N"], c["NB"]): c["rel_error"] for c in json.load(open(os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reference_vm_rel_error.json")))["values"]}
same = sum(1 for r in rows if "%.2e" % float(r["rel_error"]) == "%.2e" % float(gold[r["N"], r["NB"]]))
print(f"rel_error column identical to the reference's bench.csv (to the printed %.2e digits): {same}/{len(rows)} rows")
print("N      " + " ".join(f"NB={nb:<6d}" for nb in NBs) + "  best (GFLOP/s, median of runs 1-7)   max residual_fro")
for N in Ns:
    meds, errs = [], []
    for nb in NBs:
        g = [float(r["gflops"]) for r in rows if r["N"] == N and r["NB"] == nb and r["run_idx"] > 0]
        e = [float(r["residual_fro"]) for r in rows if r["N"] == N and r["NB"] == nb and float(r["residual_fro"]) >= 0]
        meds.append(statistics.median(g) if g else float("nan"))
        errs += e
    best = max(range(len(NBs)), key=lambda i: meds[i])
    print(f"{N:<6d} " + " ".join(f"{m:9.1f}" for m in meds) + f"  {meds[best]:9.1f} @NB={NBs[best]}   {max(errs):.2e}")
