#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/trace_one.sh rows4 4096 512 CHOLMI_FLOW_ROWS=4 > gpurun_out/rows2c_a.log 2>&1
python3 scripts/gantt.py gpurun_out/trace_rows4.csv 0 800 > gpurun_out/gantt_rows4_window.txt
