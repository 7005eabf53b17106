#!/bin/bash
# round 4: whole panel in the flow x column slices x near column
cd ${GRAFT_REPO_ROOT:-/root/repo}
CHOLMI_COL_SLICES=1 CHOLMI_PIPE_NEAR=1 timeout -k 10 300 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "flow or variants" > gpurun_out/rows3_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/rows3_pytest.log
for cfg in 4096x512 6144x512 3072x384; do
for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
echo "== $cfg COL_SLICES=$1 PIPE_NEAR=$2"; CHOLMI_COL_SLICES=$1 CHOLMI_PIPE_NEAR=$2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $cfg 2>&1 | grep "rep=2"
done; done
for v in "0 0 -1" "0 0 0.7" "1 0 0.7" "0 1 0.7" "1 1 0.7" "0 1 -1"; do set -- $v
echo "== 8192x512 COL_SLICES=$1 PIPE_NEAR=$2 FLOW_FACTOR=$3"; CHOLMI_COL_SLICES=$1 CHOLMI_PIPE_NEAR=$2 CHOLMI_FLOW_FACTOR=$3 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py 8192x512 2>&1 | grep "rep=2"
done
