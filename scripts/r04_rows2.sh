#!/bin/bash
# round 4: streaming row slabs (k_flow_rows2) against the eager ones: parity, then time, then stamps
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "flow" > gpurun_out/rows2_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/rows2_pytest.log
for cfg in 2048x512 4096x512 3072x384 6144x512; do
for f in 1 2 3 4; do
echo "== $cfg FLOW_ROWS=$f"; CHOLMI_FLOW_ROWS=$f PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $cfg 2>&1 | grep "rep=2"
done; done
for f in 4; do CHOLMI_FLOW_ROWS=$f timeout -k 10 120 python scripts/flow_marks.py 4096x512 0 900 > gpurun_out/rows2_marks_$f.txt 2>&1; CHOLMI_FLOW_ROWS=$f timeout -k 10 120 python scripts/flow_stamps.py 4096x512 > gpurun_out/rows2_stamps_$f.txt 2>&1; tail -3 gpurun_out/rows2_stamps_$f.txt; done
