#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $1 NEARF=$2 U1S=$3"; CHOLMI_NEAR_FACTOR=$2 CHOLMI_U1_SMALL=$3 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 8192x512 12288x512 16384x512 16384x512 8192x1024 16384x1024; do
for v in "0.7 8" "0.5 8" "0.35 8" "0.2 8" "0 0"; do set -- $v; run $cfg $1 $2; done; done
