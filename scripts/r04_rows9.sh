#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/trace_one.sh n8k 8192 512 > gpurun_out/rows9_a.log 2>&1
python3 scripts/gantt.py gpurun_out/trace_n8k.csv 3300 4600 > gpurun_out/gantt_n8k_window.txt
python3 scripts/gantt.py gpurun_out/trace_n8k.csv 0 1400 > gpurun_out/gantt_n8k_window0.txt
rm -rf gpurun_out/trace_n8k gpurun_out/trace_n8k.csv
