#!/bin/bash
# round 4: the column-(k+1) launch behind counters instead of events
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 400 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "flow or variants or stress or abort" > gpurun_out/rows8_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/rows8_pytest.log
run() { echo "== $1 U1C=$2"; CHOLMI_U1_COUNTERS=$2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 2048x512 4096x512 6144x512 6144x384 8192x512 12288x512 16384x512; do
for v in 1 0 1 0; do run $cfg $v; done; done
