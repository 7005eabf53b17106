#!/usr/bin/env python3
"""Every dispatch of a time window of the LAST factorisation in a rocprofv3 --kernel-trace CSV, one line each:
   python scripts/gantt.py <kernel_trace.csv> <from_us> <to_us>
start / end in us from the factorisation's first dispatch, duration, queue, grid size, kernel.  Also (no window given)
a per-wave summary: a wave starts at its k_flow_factor (or its first k_potrf_diag)."""
import csv
import sys

f = sys.argv[1]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void cholmi::", "").replace("cholmi::", "").strip()
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                 int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
rows.sort()
last_gen = max(i for i, r in enumerate(rows) if r[2].startswith("k_plgsy"))
rows = [r for r in rows[last_gen + 1:] if r[2].startswith("k_") and not r[2].startswith("k_residual")]
t0 = rows[0][0]
us = lambda t: (t - t0) / 1e3
if len(sys.argv) >= 4:
    lo, hi = float(sys.argv[2]), float(sys.argv[3])
    qs = sorted({r[3] for r in rows})
    print(f"# queues: {qs}")
    for s, e, n, q, g, wg in rows:
        if us(e) >= lo and us(s) <= hi:
            print(f"{us(s):9.1f} {us(e):9.1f} {(e - s) / 1e3:8.1f}  q{qs.index(q)}  wgs={g // max(1, wg):5d}  {n[:60]}")
    sys.exit(0)
starts = []
i = 0
while i < len(rows):
    n = rows[i][2]
    if n.startswith("k_flow_factor"):
        starts.append(rows[i][0])
    elif n.startswith("k_potrf_diag"):
        # the first of a run of diagonal-block launches of one tile: a new wave when the previous wave start is > 60 us back and
        # no k_potrf_diag ran in between (tile/128 of them per wave)
        if not starts or all(not (r[2].startswith("k_potrf_diag") and starts[-1] < r[0] < rows[i][0] and False) for r in rows):
            pass
    i += 1
print(f"# {len(rows)} dispatches, span {us(max(r[1] for r in rows)):.1f} us; {len(starts)} flow-form waves")
for k, w0 in enumerate(starts):
    w1 = starts[k + 1] if k + 1 < len(starts) else None
    ff = next(r for r in rows if r[0] == w0)
    fr = next((r for r in rows if r[2].startswith("k_flow_rows") and abs(r[0] - w0) < 200e3), None)
    upd = [(s, e) for s, e, n, *_ in rows if n.startswith("k_trail_update") and s >= w0 and (w1 is None or s < w1)]
    ub = sum(e - s for s, e in upd) / 1e3
    print(f"wave@{us(w0):8.1f}  span {'%7.1f' % ((w1 - w0) / 1e3) if w1 else '      -'}  flow_factor {(ff[1] - ff[0]) / 1e3:6.1f}  flow_rows {((fr[1] - fr[0]) / 1e3) if fr else 0:6.1f}  update launches {len(upd)} sum {ub:7.1f} us")
