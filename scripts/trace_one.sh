#!/bin/bash
# rocprofv3 kernel trace of one factorisation: bash scripts/trace_one.sh <tag> <N> <tile> [env assignments...]
set -u
tag=$1; N=$2; B=$3; shift 3
root=${GRAFT_REPO_ROOT:-/root/repo}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/trace_$tag
rm -rf "$out"; mkdir -p "$out"
PROBE_QUICK=1 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$root/scripts/probe_perf.py" ${N}x${B} ${N}x${B} > "$root/gpurun_out/trace_$tag.log" 2>&1
cd "$root"
t=$(find "$out" -name "*kernel_trace.csv" | head -1)
cp "$t" "gpurun_out/trace_${tag}.csv"
python3 scripts/gantt.py "gpurun_out/trace_${tag}.csv" > "gpurun_out/gantt_${tag}.txt"
tail -3 "gpurun_out/trace_$tag.log"
cat "gpurun_out/gantt_${tag}.txt"
