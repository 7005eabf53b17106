#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>" <probe args...>    (one rocprofv3 --pmc pass; run on the GPU box)
set -u
tag=$1; shift
ctrs=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --output-format csv -d "$out" -- python3 "$root/scripts/probe_perf.py" "$@" > "$root/gpurun_out/pmc_$tag.log" 2>&1
cd "$root"
f=$(find "$out" -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
rows = list(csv.DictReader(open(f)))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
disp = collections.defaultdict(set)
for r in rows: disp[r["Kernel_Name"].split("(")[0][-40:]].add(r["Dispatch_Id"])
with open(f"gpurun_out/pmc_{tag}_summary.txt", "w") as o:
    for k, d in agg.items():
        line = f"{k} dispatches={len(disp[k])} " + " ".join(f"{c}={v:.6g}" for c, v in sorted(d.items()))
        print(line); o.write(line + "\n")
PY
else echo "no counter file"; tail -5 "$root/gpurun_out/pmc_$tag.log"; fi
