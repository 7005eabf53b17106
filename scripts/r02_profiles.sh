#!/bin/bash
# round-2 evidence pass (GPU box): numbers table, timelines, PMC for the shipped kernels, issue time
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; cd "$root"; mkdir -p gpurun_out
python scripts/dist_issue_time.py > gpurun_out/r02_dist_issue_time.txt 2>&1; tail -6 gpurun_out/r02_dist_issue_time.txt
bash scripts/trace_cfg.sh r02after16k 16384x512 > /dev/null 2>&1; head -3 gpurun_out/timeline_r02after16k.txt
bash scripts/trace_cfg.sh r02after32k 32768x512 > /dev/null 2>&1; head -3 gpurun_out/timeline_r02after32k.txt
bash scripts/pmc_bench.sh r02b_headline 65536 1024 > gpurun_out/r02b_pmc_headline.log 2>&1; grep "k_trail_update" gpurun_out/pmc_r02b_headline_summary.txt | cut -c1-400
bash scripts/pmc_bench.sh r02b_config3 32768 512 > gpurun_out/r02b_pmc_config3.log 2>&1; grep "k_trail_update" gpurun_out/pmc_r02b_config3_summary.txt | cut -c1-400
bash scripts/pmc_bench.sh r02b_config2 16384 512 > gpurun_out/r02b_pmc_config2.log 2>&1; grep "k_trail_update" gpurun_out/pmc_r02b_config2_summary.txt | cut -c1-400
bash scripts/round_numbers.sh > gpurun_out/r02_numbers.log 2>&1; cat gpurun_out/numbers.txt
