#!/usr/bin/env python3
"""Per kernel of a rocprofv3 --kernel-trace CSV: calls, sum of dispatch durations, and the UNION of
their intervals (wall time during which at least one dispatch of that kernel runs).  Launches of the
trailing update run side by side (column launches beside the big one), so the sum
of their durations exceeds the wall time they occupy; the union is what bench.py's HIP-event brackets
measure and what its `roofline.achieved` divides by.
    python scripts/union_busy.py <kernel_trace.csv> [flops_per_factorisation n_factorisations]"""
import csv
import sys

rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void cholmi::", "").strip()
    rows.setdefault(n, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))


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


print("kernel                               calls   sum_of_durations_ms   union_ms   sum/union")
upd = []
for n, iv in sorted(rows.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    sd, un = sum(e - s for s, e in iv) / 1e6, union(iv) / 1e6
    print(f"{n[:36]:36s} {len(iv):6d} {sd:18.3f} {un:12.3f} {sd / max(un, 1e-9):8.3f}")
    if n.startswith("k_trail_update"):
        upd += iv
if upd:
    un = union(upd) / 1e6
    print(f"# all k_trail_update* dispatches: {len(upd)} calls, sum {sum(e - s for s, e in upd) / 1e6:.3f} ms, union {un:.3f} ms")
    if len(sys.argv) > 3:
        fl, nf = float(sys.argv[2]), int(sys.argv[3])
        print(f"# update flops {fl:.4e} per factorisation x {nf}: {fl * nf / (un * 1e-3) / 1e12:.2f} TFLOP/s over the union")
