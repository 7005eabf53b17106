#!/bin/bash
# kernel trace of one factorisation (GPU box): scripts/trace_cfg.sh <tag> <N>x<tile>   (env switches pass through)
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; cfg=$2
o=$root/gpurun_out/tl_$tag; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
PROBE_QUICK=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o -- python3 $root/scripts/probe_perf.py $cfg $cfg > $o.log 2>&1
f=$(find $o -name "*kernel_trace.csv" | head -1)
cd $root
[ -n "$f" ] && python3 scripts/timeline.py $f ${cfg#*x} > gpurun_out/timeline_$tag.txt && python3 scripts/trace_dump.py $f > gpurun_out/dump_$tag.txt
head -6 gpurun_out/timeline_$tag.txt
