#!/usr/bin/env python3
"""Dispatch list (start offset, duration, kernel, workgroups) of the last factorisation in a kernel trace."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void cholmi::", "").strip()
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, g // max(1, wg)))
rows.sort()
lg = max(i for i, r in enumerate(rows) if r[2].startswith("k_plgsy"))
rows = rows[lg + 1:]
t0 = rows[0][0]
for s, e, n, g in rows:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {n[:30]:30s} wgs={g}")
