#!/bin/bash
# usage: scripts/prof.sh <tag> <script args...>   (run on the GPU box via gpurun)
# rocprofv3 kernel trace + stats of scripts/probe_perf.py; summary -> gpurun_out/<tag>_kernel_stats.csv
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/scripts/probe_perf.py" "$@" > "$root/gpurun_out/prof_$tag.log" 2>&1
cd "$root"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "gpurun_out/${tag}_kernel_stats.csv"; cut -c1-200 "gpurun_out/${tag}_kernel_stats.csv"; else echo "no stats file"; tail -5 "gpurun_out/prof_$tag.log"; fi
