#!/bin/bash
# round 3, first call: the distributed loop on a 1 x 1 descriptor against the walker, same box (before the unification)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r03_before.txt
: > $out
run() { python bench.py --no-cpu-baseline --no-check "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['config']['N'], d['config']['tile'], d['dtype'], d['value'], 'TF/s', d['ms_per_step'], 'ms')" >> $out; }
for cfg in "--N 65536 --tile 1024 --steps 3" "--N 32768 --tile 512 --steps 3" "--N 16384 --tile 512 --steps 5" "--N 8192 --tile 512 --steps 8"; do
  echo -n "walker " >> $out; run $cfg
  echo -n "dist1x1 " >> $out; CHOLMI_FORCE_DIST=1 run $cfg
  echo -n "dist1x1-nolookahead " >> $out; CHOLMI_FORCE_DIST=1 CHOLMI_DIST_LOOKAHEAD=0 run $cfg
done
cat $out
