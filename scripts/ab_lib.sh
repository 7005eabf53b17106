#!/bin/bash
# same-box A/B of two builds of the library (GPU box): scripts/ab_lib.sh <base.so> "<cfg> <cfg> ..."   cfg = NxB
# base.so = an earlier build kept under scripts/exp/_ab/ (git-ignored, travels with the snapshot)
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; cd "$root"
base=$1; shift
for cfg in $@; do
  for v in base new base new; do
    if [ $v = base ]; then export LIBCHOLMI_PATH=$root/$base; else unset LIBCHOLMI_PATH; fi
    echo "$v $(python scripts/probe_perf.py $cfg 2>/dev/null | grep 'rep=2' | sed 's/host_issue.*//')"
  done
done
