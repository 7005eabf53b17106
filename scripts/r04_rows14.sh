#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/trace_one.sh n16k 16384 512 CHOLMI_CALIB=72.8,51.2,148.7,51.8 > gpurun_out/rows14_a.log 2>&1
python3 scripts/wave_table.py gpurun_out/trace_n16k.csv 16384 512 > gpurun_out/wave_table_16384x512.txt
python3 scripts/gantt.py gpurun_out/trace_n16k.csv 20000 22500 > gpurun_out/gantt_n16k_window.txt
rm -rf gpurun_out/trace_n16k gpurun_out/trace_n16k.csv
