#!/bin/bash
# round 4: how far up the counter-linked regime pays now that column k+2 has a launch of its own
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $1 PIPEF=$2 NEARF=$3 FLOW=$4"; CHOLMI_PIPE_FACTOR=$2 CHOLMI_NEAR_FACTOR=$3 CHOLMI_FLOW=$4 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 8192x512 12288x512 16384x512 32768x512 16384x1024; do
for v in "0.7 0.7 0" "1.0 1.0 0" "1.5 1.5 0" "2.0 2.0 0" "3.0 3.0 0" "1.5 0.7 0" "0.7 0.7 1"; do set -- $v; run $cfg $1 $2 $3; done; done
