#!/usr/bin/env python3
"""Per-wave table of the LAST factorisation in a rocprofv3 --kernel-trace CSV: python scripts/wave_table.py <kernel_trace.csv> <N> <tile>
A wave starts at its k_flow_factor or at the first of its tile/128 k_potrf_diag launches.  Columns: start, span to the next wave's
start, union of the k_trail_update launches that started inside the wave, what the wave's update would take at 70 TFLOP/s."""
import csv, sys
f, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nbm, nt = B // 128, N // B
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void cholmi::", "").replace("cholmi::", "").strip()
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
last_gen = max(i for i, r in enumerate(rows) if r[2].startswith("k_plgsy"))
rows = [r for r in rows[last_gen + 1:] if r[2].startswith("k_") and not r[2].startswith("k_residual")]
t0 = rows[0][0]
starts, nd = [], 0
for s, e, n in rows:
    if n.startswith("k_flow_factor"):
        starts.append((s, "flow"))
    elif n.startswith("k_potrf_diag"):
        if nd % nbm == 0:
            starts.append((s, "step"))
        nd += 1
print(f"# N={N} tile={B}: {len(rows)} dispatches, span {(max(r[1] for r in rows) - t0) / 1e3:.1f} us, {len(starts)} waves found (nt = {nt})")
print("# wave  form   start_us   span_us  update_union_us  update_at_70TF_us  update launches")
for k, (w0, form) in enumerate(starts):
    w1 = starts[k + 1][0] if k + 1 < len(starts) else max(r[1] for r in rows)
    iv = sorted((s, e) for s, e, n in rows if n.startswith("k_trail_update") and w0 <= s < w1)
    un, hi = 0, -1
    for s, e in iv:
        if s > hi:
            un += e - s
        elif e > hi:
            un += e - hi
        hi = max(hi, e)
    m = nt - 1 - k  # tiles below the diagonal in column k
    flops = (m * (m - 1) / 2 * 2 + m) * B ** 3  # GEMMs 2 B^3, SYRKs B^3
    print(f"{k:5d}  {form}  {(w0 - t0) / 1e3:9.1f} {(w1 - w0) / 1e3:9.1f} {un / 1e3:12.1f} {flops / 70e12 * 1e6:14.1f} {len(iv):8d}")
