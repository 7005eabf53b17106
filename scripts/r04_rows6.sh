#!/bin/bash
# round 4: new defaults (near column, latency-form column update up to 8 tiles): thresholds and the flow rule again
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $1 FLOW=$2 U1S=$3 NEAR=$4"; CHOLMI_FLOW=$2 CHOLMI_U1_SMALL=$3 CHOLMI_PIPE_NEAR=$4 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 2048x512 3072x384 4096x512 6144x512 6144x384; do
for v in "1 8 1" "0 8 1" "1 16 1" "1 4 1" "1 0 0"; do set -- $v; run $cfg $1 $2 $3; done; done
for cfg in 8192x512 12288x512 16384x512 8192x1024 16384x1024 32768x512; do
for v in "1 8 1" "1 12 1" "1 4 1" "1 0 0"; do set -- $v; run $cfg $1 $2 $3; done; done
