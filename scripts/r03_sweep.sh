#!/bin/bash
# regime thresholds re-checked against the measured calibration (units changed slightly in round 3)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r03_sweep.txt
: > $out
for cfg in 16384:512:6 32768:512:3 8192:512:10; do
  bash scripts/ab.sh "CHOLMI_X=0 CHOLMI_PIPE_FACTOR=0.5 CHOLMI_PIPE_FACTOR=1.0 CHOLMI_PIPE_FACTOR=1.5 CHOLMI_YIELD_FACTOR=2 CHOLMI_YIELD_FACTOR=4 CHOLMI_PAIR_FACTOR=1.5 CHOLMI_PAIR_FACTOR=3 CHOLMI_HALVES_MAX_ROUNDS=12 CHOLMI_HALVES_MAX_ROUNDS=48 CHOLMI_X=0" $cfg >> $out 2>&1
done
cat $out
