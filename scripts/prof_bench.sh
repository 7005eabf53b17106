#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command itself (GPU box): bash scripts/prof_bench.sh <tag> [bench args]
# -> gpurun_out/bench_<tag>_kernel_stats.csv (rocprof's own summary), bench_<tag>_union.txt (scripts/union_busy.py),
#    bench_<tag>_line.json (the line that run printed)
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/benchprof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-worker-path --no-live-traffic "$@" > "$root/gpurun_out/benchprof_$tag.log" 2>&1
cd "$root"
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "gpurun_out/bench_${tag}_kernel_stats.csv"
grep "\"metric\"" "gpurun_out/benchprof_$tag.log" | tail -1 > "gpurun_out/bench_${tag}_line.json"
t=$(find "$out/trace" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 scripts/union_busy.py "$t" > "gpurun_out/bench_${tag}_union.txt"
cut -c1-150 "gpurun_out/bench_${tag}_kernel_stats.csv" | head -6
cat "gpurun_out/bench_${tag}_union.txt" | head -12
python3 -c "import json; d=json.load(open('gpurun_out/bench_${tag}_line.json')); print(d['value'], d['roofline'])"
