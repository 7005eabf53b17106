#!/bin/bash
# round 4: fused in-tile steps in event-linked yield waves
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 500 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "flow or variants or config2 or abort" > gpurun_out/rows15_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/rows15_pytest.log
run() { echo "== $1 $2"; env $2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 8192x512 12288x512 16384x512 32768x512 16384x1024 32768x1024 65536x1024; do
for v in "X=1" "CHOLMI_FUSE_PLAIN=0" "X=1" "CHOLMI_FUSE_PLAIN=0"; do run $cfg $v; done; done
