#!/bin/bash
# round 4: where the whole-factorisation flow rule should end
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $1 $2"; env $2 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $1 2>&1 | grep "rep=2" | cut -c1-75; }
for cfg in 7168x512 8192x512 9216x512 10240x512 12288x512 8192x384 9216x384; do
for v in "CHOLMI_FLOW_RUN_FACTOR=100" "CHOLMI_FLOW=0" "CHOLMI_FLOW_RUN_FACTOR=100" "CHOLMI_FLOW=0" "X=1"; do run $cfg $v; done; done
bash scripts/round_numbers.sh > gpurun_out/numbers2.log 2>&1; cat gpurun_out/numbers.txt
