#!/bin/bash
# round 4: the flow form entered late (per-wave rule) with the final schedule
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $1 $2"; env $2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 10240x512 12288x512 16384x512; do
for v in "X=1" "CHOLMI_FLOW_FACTOR=0.3" "CHOLMI_FLOW_FACTOR=0.5" "CHOLMI_FLOW_FACTOR=0.7" "X=1" "CHOLMI_FLOW=0"; do run $cfg "$v"; done; done
