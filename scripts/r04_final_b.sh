#!/bin/bash
# round 4, final measurements, part B: the table of configurations, the PMC passes at the headline size, the task path
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
out=gpurun_out/r04_final_b.txt
: > $out
echo "--- table" >> $out
bash scripts/round_numbers.sh >> $out 2>&1
echo "--- pmc" >> $out
bash scripts/pmc_bench.sh r04_headline 65536 1024 >> $out 2>&1
echo "--- worker path" >> $out
python scripts/worker_profile.py 16384 512 2>&1 | grep "^device_results" >> $out
python scripts/worker_profile.py 8192 512 2>&1 | grep "^device_results" >> $out
tail -70 $out
