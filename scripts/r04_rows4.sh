#!/bin/bash
# round 4: same-box comparison of the row-slab forms and the near column
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $1 ROWS=$2 NEAR=$3 FF=$4"; CHOLMI_FLOW_ROWS=$2 CHOLMI_PIPE_NEAR=$3 CHOLMI_FLOW_FACTOR=$4 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 4096x512 6144x512; do
for v in "1 0" "2 0" "2 1" "3 1" "4 1" "1 1"; do set -- $v; run $cfg $1 $2 -1; done; done
for cfg in 8192x512 16384x512; do
for v in "1 0 -1" "1 1 -1" "2 1 0.7" "4 1 0.7" "2 1 0.4" "4 1 0.4" "2 0 0.7"; do set -- $v; run $cfg $1 $2 $3; done; done
