#!/bin/bash
# A/B of environment switches on the bench line (GPU box): scripts/ab.sh "<VAR=a VAR=b ...>" cfg...   cfg = N:tile[:steps]
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}; cd "$root"
variants=$1; shift
for cfg in "$@"; do
  IFS=: read N T S <<< "$cfg"; S=${S:-3}
  for v in $variants; do
    line=$(env $v python bench.py --no-cpu-baseline --no-live-traffic --N $N --tile $T --steps $S --warmup 1 2>/dev/null | tail -1)
    echo "$v N=$N tile=$T: $(echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d.get('roofline',{}); print(d['value'],'TF/s', d['pct_of_mfma_peak'],'%  upd', r.get('achieved'), 'res', d.get('residual'))")"
  done
done
