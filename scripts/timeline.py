#!/usr/bin/env python3
"""Per-wave timeline of one factorisation from a rocprofv3 --kernel-trace CSV:
   python scripts/timeline.py <kernel_trace.csv> <tile> [out.txt]
The last factorisation in the trace is cut at the last k_plgsy dispatch.  A wave starts at the
first diagonal-block launch (k_potrf_diag) of its POTRF; tile/128 such launches per wave.
Per wave: panel chain (first k_potrf_diag start .. last panel-kernel end), the busy time of the
trailing update (union of k_trail_update / diagonal SYRK intervals), the time in the wave in
which NO update kernel runs (gap), and the tail of every update launch (end of launch minus the
time at which its average block rate would have finished: not observable from a trace, so the
gap is what is reported)."""
import csv
import sys

f, tile = sys.argv[1], int(sys.argv[2])
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void cholmi::", "").strip()
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
last_gen = max(i for i, r in enumerate(rows) if r[2].startswith("k_plgsy"))
rows = [r for r in rows[last_gen + 1:] if r[2].startswith("k_") and not r[2].startswith("k_residual")]
t0 = rows[0][0]
tend = max(r[1] for r in rows)
nbm = tile // 128
potrf = [r for r in rows if r[2].startswith("k_potrf_diag")]
waves = [potrf[i][0] for i in range(0, len(potrf), nbm)]
upd = lambda n: n.startswith("k_trail_update")
pan = lambda n: not upd(n)


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
    if cs is not None:
        tot += ce - cs
    return tot


print(f"# {f}: {len(rows)} dispatches, {len(waves)} waves, span {(tend - t0) / 1e6:.3f} ms")
ub = union([(s, e) for s, e, n in rows if upd(n)])
pb = union([(s, e) for s, e, n in rows if pan(n)])
both = union([(s, e) for s, e, n in rows])
print(f"# update busy (union) {ub / 1e6:.3f} ms = {100 * ub / (tend - t0):.1f} % of span; panel-chain kernels busy {pb / 1e6:.3f} ms; "
      f"nothing running {(tend - t0 - both) / 1e6:.3f} ms")
print("# wave  span_us  upd_busy_us  upd_gap_us  panel_busy_us  n_upd_launches  n_panel_launches")
tot_gap = 0
for k, w0 in enumerate(waves):
    w1 = waves[k + 1] if k + 1 < len(waves) else tend
    clip = lambda s, e: (max(s, w0), min(e, w1))
    u = [clip(s, e) for s, e, n in rows if upd(n) and e > w0 and s < w1]
    p = [clip(s, e) for s, e, n in rows if pan(n) and e > w0 and s < w1]
    ubk, pbk = union(u), union(p)
    nu = sum(1 for s, e, n in rows if upd(n) and w0 <= s < w1)
    np_ = sum(1 for s, e, n in rows if pan(n) and w0 <= s < w1)
    tot_gap += (w1 - w0) - ubk
    if True:
        print(f"{k:4d} {(w1 - w0) / 1e3:9.1f} {ubk / 1e3:9.1f} {(w1 - w0 - ubk) / 1e3:9.1f} {pbk / 1e3:9.1f} {nu:4d} {np_:4d}")
print(f"# sum over waves of time without an update kernel running: {tot_gap / 1e6:.3f} ms of {(tend - t0) / 1e6:.3f} ms")
# launch-level: duration of every update launch vs its share of flops is not in the trace; report the
# distribution of gaps between consecutive update launches instead
us = sorted((s, e) for s, e, n in rows if upd(n))
gaps = [max(0, us[i + 1][0] - max(x[1] for x in us[: i + 1])) for i in range(len(us) - 1)]
if gaps:
    gaps.sort()
    print(f"# gaps between consecutive update launches: median {gaps[len(gaps) // 2] / 1e3:.1f} us, "
          f"p90 {gaps[int(0.9 * len(gaps))] / 1e3:.1f} us, max {gaps[-1] / 1e3:.1f} us, total {sum(gaps) / 1e6:.3f} ms")
