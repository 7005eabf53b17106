#!/bin/bash
# round 4: the near column in the latency form
cd ${GRAFT_REPO_ROOT:-/root/repo}
CHOLMI_NEAR_SMALL=64 timeout -k 10 500 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "flow or variants" > gpurun_out/rows12_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/rows12_pytest.log
run() { echo "== $1 $2"; env $2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 2048x512 4096x512 6144x512 6144x384 8192x512 12288x512 16384x512; do
for v in "X=1" "CHOLMI_NEAR_SMALL=8" "CHOLMI_NEAR_SMALL=4" "X=1" "CHOLMI_NEAR_SMALL=8" "CHOLMI_NEAR_SMALL=12"; do run $cfg $v; done; done
