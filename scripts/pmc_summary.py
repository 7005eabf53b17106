#!/usr/bin/env python3
"""Reduce the passes of scripts/pmc_bench.sh to one table per kernel: MFMA-busy %, issue/wait
split, LDS bank-conflict share, L2 hit rate, HBM-side bytes (FETCH_SIZE doubled: the gfx950
correction of MI355X_MICROARCH.md, section HBM) and GB/s over the kernel-trace durations."""
import collections
import csv
import glob
import json
import sys

out, tag, N, tile, dtype = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]


def short(name):
    n = name.split("(")[0]
    return n.replace("void cholmi::", "").strip()


ctr = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(f"{out}/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add((f, r["Dispatch_Id"]))
dur = collections.defaultdict(float)
calls = collections.Counter()
for f in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        calls[k] += 1
print(f"# rocprofv3 counters over ONE factorisation, bench.py --N {N} --tile {tile} --dtype {dtype} (tag {tag})")
print("# MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); clock = GRBM_GUI_ACTIVE/8 / kernel time (PMC pass)")
print("# HBM bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH doubled per the gfx950 correction); GB/s over the un-profiled kernel-trace durations")
res = {}
for k in sorted(ctr, key=lambda k: -dur.get(k, 0)):
    c = ctr[k]
    if dur.get(k, 0) <= 0:
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8
    row = {"calls": calls[k], "total_ms": dur[k] * 1e3}
    if gui:
        row["mfma_busy_pct"] = 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024)
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        row["wave_cycles_active_pct"] = 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc
        row["wave_cycles_wait_any_pct"] = 100 * c.get("SQ_WAIT_ANY", 0) / wc
        row["wave_cycles_wait_inst_pct"] = 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc
        row["wave_cycles_wait_inst_lds_pct"] = 100 * c.get("SQ_WAIT_INST_LDS", 0) / wc
    if c.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_pct"] = 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        row["l2_hit_pct"] = 100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)
    if mops:
        row["mfma_mops"] = mops
    hbm = 2 * c.get("FETCH_SIZE", 0) * 1024 + c.get("WRITE_SIZE", 0) * 1024
    if hbm:
        row["hbm_bytes"] = hbm
        row["hbm_GBps"] = hbm / dur[k] / 1e9
        row["hbm_bytes_per_launch"] = hbm / max(1, calls[k])
    res[k] = row
    print(k, " ".join(f"{a}={b:.4g}" if isinstance(b, float) else f"{a}={b}" for a, b in row.items()))
json.dump({"N": N, "tile": tile, "dtype": dtype, "kernels": res}, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
