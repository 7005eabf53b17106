#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command itself + PMC passes for HBM traffic of the
# dominant kernel.  Run on the GPU box: bash scripts/prof_bench.sh <tag>
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/benchprof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$root/gpurun_out/benchprof_$tag.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$root/gpurun_out/bench_${tag}_kernel_stats.csv"
grep "\"metric\"" "$root/gpurun_out/benchprof_$tag.log" | tail -1 > "$root/gpurun_out/bench_${tag}_line.json"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$root/gpurun_out/benchpmc_${tag}_$c.log" 2>&1
done
cd "$root"
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)
    tot = 0.0; disp = set()
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "k_trail_update" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    res[c] = (tot, len(disp))
# rocprofv3 reports both in KiB; gfx950 FETCH_SIZE counts wide coalesced reads at half (MI355X_MICROARCH.md HBM)
fetch_b = res["FETCH_SIZE"][0] * 1024 * 2
write_b = res["WRITE_SIZE"][0] * 1024
n = max(1, res["FETCH_SIZE"][1])
j = {"N": 65536, "tile": 1024, "kernel": "k_trail_update", "launches": n,
     "fetch_bytes_per_launch_corrected_x2": fetch_b / n, "write_bytes_per_launch": write_b / max(1, res["WRITE_SIZE"][1]),
     "hbm_bytes_per_launch": fetch_b / n + write_b / max(1, res["WRITE_SIZE"][1]),
     "hbm_bytes_per_factorisation": fetch_b + write_b,
     "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over one factorisation; FETCH_SIZE doubled per the gfx950 correction"}
json.dump(j, open(f"gpurun_out/pmc_traffic_{tag}.json", "w"), indent=1)
print(json.dumps(j))
PY
cut -c1-160 gpurun_out/bench_${tag}_kernel_stats.csv | head -8
cat gpurun_out/bench_${tag}_line.json
