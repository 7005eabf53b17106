#!/bin/bash
# round-2 first GPU pass: new parity tests, bench line, timelines at configs 2/3, PMC at headline + config 3
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
cd "$root"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_conditioning.py tests/test_gpu_abi.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r02_t1.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02_t1.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench_a.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r02_bench_a.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
for cfg in 16384x512 32768x512; do
  o=$root/gpurun_out/r02_tl_$cfg; mkdir -p $o
  PROBE_QUICK=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o -- python3 $root/scripts/probe_perf.py $cfg $cfg > $o.log 2>&1
  f=$(find $o -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python3 $root/scripts/timeline.py $f ${cfg#*x} > $root/gpurun_out/r02_timeline_$cfg.txt && head -14 $root/gpurun_out/r02_timeline_$cfg.txt
done
cd "$root"
