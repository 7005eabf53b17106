#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/trace_one.sh n4k 4096 512 > gpurun_out/rows9_a.log 2>&1
python3 scripts/gantt.py gpurun_out/trace_n4k.csv 0 1300 > gpurun_out/gantt_n4k_window.txt
rm -rf gpurun_out/trace_n4k gpurun_out/trace_n4k.csv
