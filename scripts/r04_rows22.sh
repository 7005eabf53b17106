#!/bin/bash
# round 4: the two-columns-ahead defaults at other tile sizes (tuned at 512)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 400 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "flow or variants or config2" > gpurun_out/rows22_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/rows22_pytest.log
run() { echo "== $1 $2"; env $2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 8192x1024 12288x1024 16384x1024 65536x1024; do
for v in "X=1" "CHOLMI_U1_SMALL=0" "X=1" "CHOLMI_U1_SMALL=0"; do run $cfg "$v"; done; done
