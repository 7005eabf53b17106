#!/bin/bash
# round 4: the yield threshold again (the update's waves sleep while a panel-chain workgroup is on their CU)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $1 $2"; env $2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 12288x512 16384x512 32768x512 16384x1024; do
for v in "X=1" "CHOLMI_YIELD_FACTOR=2" "CHOLMI_YIELD_FACTOR=1.3" "CHOLMI_YIELD_FACTOR=0.9" "CHOLMI_YIELD_FACTOR=0.71" "X=1" "CHOLMI_YIELD_FACTOR=5"; do run $cfg $v; done; done
