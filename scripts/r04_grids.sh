#!/bin/bash
# round 4: the counter-linked POTRF -> TRSM edge on grids -- rehearsal tests with the form forced, rank-alone device times on / off
cd ${GRAFT_REPO_ROOT:-/root/repo}
CHOLMI_PIPE_FACTOR=100 CHOLMI_PAIR_FACTOR=1000 timeout -k 10 500 python -m pytest tests/test_gpu_distributed.py -x -q > gpurun_out/r04_grid_forced_pytest.log 2>&1; echo "forced rc=$?"; tail -3 gpurun_out/r04_grid_forced_pytest.log
timeout -k 10 300 python -m pytest tests/test_gpu_distributed.py -x -q > gpurun_out/r04_grid_default_pytest.log 2>&1; echo "default rc=$?"; tail -2 gpurun_out/r04_grid_default_pytest.log
for cfg in "65536 1024" "16384 512"; do
  set -- $cfg
  echo "## N=$1 tile=$2, counters on the local edge (default)"; timeout -k 10 250 python scripts/dist_issue_time.py $1 $2 2>/dev/null | grep -E "grid (4x2|2x4|8x1)"
  echo "## N=$1 tile=$2, CHOLMI_DEVICE_FLAGS=0 (events only, as in round 3)"; CHOLMI_DEVICE_FLAGS=0 timeout -k 10 250 python scripts/dist_issue_time.py $1 $2 2>/dev/null | grep -E "grid (4x2|2x4|8x1)"
done > gpurun_out/r04_rank_alone.txt
python3 - <<'PY'
import re,collections
cur=None; d=collections.OrderedDict()
for l in open('gpurun_out/r04_rank_alone.txt'):
    if l.startswith('##'): cur=l.strip('# \n'); continue
    m=re.match(r'grid (\S+) rank (\d+):.*schedule ([\d.]+) ms',l)
    if m: d.setdefault((cur,m.group(1)),[]).append(float(m.group(3)))
for (c,g),v in d.items(): print(f"{c:70s} grid {g}: slowest rank {max(v):8.2f} ms, fastest {min(v):8.2f} ms")
PY
