#!/bin/bash
# round 4, final measurements, part A: the whole GPU suite, the default bench line, rocprof of the bench command
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
out=gpurun_out/r04_final_a.txt
: > $out
timeout -k 10 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r04_final_pytest.log 2>&1
echo "pytest rc=$?" >> $out
tail -3 gpurun_out/r04_final_pytest.log >> $out
timeout -k 10 300 python bench.py > gpurun_out/r04_bench_default_line.json 2> gpurun_out/r04_bench_default.err
echo "bench rc=$?" >> $out
cat gpurun_out/r04_bench_default_line.json >> $out
bash scripts/prof_bench.sh r04 >> $out 2>&1
tail -40 $out
