#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in 1 2 3; do CHOLMI_FLOW_ROWS=$f timeout -k 10 120 python scripts/flow_marks.py 4096x512 0 900 > gpurun_out/rows2_marks_$f.txt 2>&1; CHOLMI_FLOW_ROWS=$f timeout -k 10 120 python scripts/flow_stamps.py 4096x512 > gpurun_out/rows2_stamps_$f.txt 2>&1; tail -3 gpurun_out/rows2_stamps_$f.txt; done
