#!/bin/bash
# round 4: update launches with ONE workgroup per CU in the waves that yield to the panel chain
cd ${GRAFT_REPO_ROOT:-/root/repo}
CHOLMI_OCC1_FACTOR=100 timeout -k 10 300 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "variants" > gpurun_out/rows19_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/rows19_pytest.log
run() { echo "== $1 $2"; env $2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 6144x512 8192x512 12288x512 16384x512 32768x512 16384x1024; do
for v in "X=1" "CHOLMI_OCC1_FACTOR=0.7" "CHOLMI_OCC1_FACTOR=1.5" "CHOLMI_OCC1_FACTOR=3" "X=1" "CHOLMI_OCC1_FACTOR=1"; do run $cfg $v; done; done
