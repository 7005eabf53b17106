#!/bin/bash
# round 4: column k+1 in the latency form (CHOLMI_U1_SMALL) x near column x row-slab form
cd ${GRAFT_REPO_ROOT:-/root/repo}
CHOLMI_U1_SMALL=64 CHOLMI_PIPE_NEAR=1 timeout -k 10 300 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "flow or variants" > gpurun_out/rows5_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/rows5_pytest.log
run() { echo "== $1 ROWS=$2 NEAR=$3 FF=$4 U1S=$5"; CHOLMI_FLOW_ROWS=$2 CHOLMI_PIPE_NEAR=$3 CHOLMI_FLOW_FACTOR=$4 CHOLMI_U1_SMALL=$5 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 4096x512 6144x512; do
for v in "1 0 0" "1 1 0" "1 1 64" "2 1 64" "4 1 64" "1 0 64"; do set -- $v; run $cfg $1 $2 -1 $3; done; done
for cfg in 8192x512 16384x512; do
for v in "1 0 -1 0" "1 1 -1 0" "1 1 -1 64" "1 1 -1 8" "2 1 0.7 64" "4 1 0.7 64" "2 1 0.4 64"; do set -- $v; run $cfg $1 $2 $3 $4; done; done
