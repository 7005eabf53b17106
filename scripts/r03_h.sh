#!/bin/bash
# round 3: the whole GPU suite, the default bench line, rocprof of the bench command, the table of configurations, PMC passes
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
out=gpurun_out/r03_h.txt
: > $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -v -p no:cacheprovider --timeout 500 > gpurun_out/r03_h_pytest.log 2>&1
echo "pytest rc=$?" >> $out
tail -3 gpurun_out/r03_h_pytest.log >> $out
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_default_line.json 2> gpurun_out/r03_bench_default.err
echo "bench rc=$?" >> $out
cat gpurun_out/r03_bench_default_line.json >> $out
bash scripts/prof_bench.sh r03 >> $out 2>&1
echo "--- table" >> $out
bash scripts/round_numbers.sh >> $out 2>&1
echo "--- pmc" >> $out
bash scripts/pmc_bench.sh r03_headline 65536 1024 >> $out 2>&1
tail -60 $out
