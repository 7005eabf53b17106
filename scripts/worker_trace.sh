#!/bin/bash
# kernel trace of one task-API run (GPU box): scripts/worker_trace.sh <tag> <N> <tile> [dump]
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; N=$2; B=$3
o=$root/gpurun_out/wt_$tag; rm -rf $o; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $o -- python3 $root/scripts/worker_trace.py $N $B > $o.log 2>&1
f=$(find $o -name "*kernel_trace.csv" | head -1)
cd $root
tail -2 $o.log
[ -n "$f" ] && python3 scripts/worker_trace.py --summary $f ${4:-} > gpurun_out/worker_trace_$tag.txt && head -30 gpurun_out/worker_trace_$tag.txt
