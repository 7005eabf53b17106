#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/trace_one.sh n8k 8192 512 CHOLMI_CALIB=72.8,51.2,148.7,51.8 > gpurun_out/rows18_a.log 2>&1
python3 scripts/wave_table.py gpurun_out/trace_n8k.csv 8192 512 > gpurun_out/wave_table_8192x512.txt
python3 scripts/gantt.py gpurun_out/trace_n8k.csv 0 1500 > gpurun_out/gantt_n8k_window.txt
rm -rf gpurun_out/trace_n8k gpurun_out/trace_n8k.csv
bash scripts/r04_stress.sh > gpurun_out/r04_stress_two.log 2>&1; tail -4 gpurun_out/r04_stress_two.log
