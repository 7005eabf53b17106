#!/bin/bash
# deep lookahead (CHOLMI_LOOKAHEAD = d): correctness of the forced variants, then A/B on the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
out=gpurun_out/r03_la.txt
: > $out
timeout -k 10 900 python -m pytest tests/test_gpu_full.py -m gpu -x -q -p no:cacheprovider -k "schedule_variants" --timeout 600 > gpurun_out/r03_la_pytest.log 2>&1
echo "pytest rc=$?" >> $out; tail -2 gpurun_out/r03_la_pytest.log >> $out
bash scripts/ab.sh "CHOLMI_LOOKAHEAD=0 CHOLMI_LOOKAHEAD=1 CHOLMI_LOOKAHEAD=2 CHOLMI_LOOKAHEAD=3 CHOLMI_LOOKAHEAD=0 CHOLMI_LOOKAHEAD=3" 8192:512:10 16384:512:6 4096:512:10 32768:512:3 16384:1024:5 >> $out 2>&1
bash scripts/ab.sh "CHOLMI_LOOKAHEAD=0 CHOLMI_LOOKAHEAD=3" 65536:1024:2 >> $out 2>&1
cat $out
