#!/bin/bash
# The table of measured configurations quoted in DESIGN.md / README.md (run on the GPU box).
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
cd "$root"
out=gpurun_out/numbers.txt
: > $out
run() { python bench.py --no-cpu-baseline --no-worker-path --no-live-traffic "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d.get('roofline',{}); n=d['config']['N']; print(n, d['config']['tile'], d['dtype'], d['value'], 'TF/s', d['pct_of_mfma_peak'], '% of peak; update kernel', r.get('achieved'), 'steps', d['steps'], '; one more step without the event brackets:', round(n**3/3/(d['unprofiled_ms']*1e-3)/1e12, 2), 'TF/s')" >> $out; }
run --N 65536 --tile 1024
run --N 32768 --tile 1024
run --N 32768 --tile 512
run --N 16384 --tile 1024 --steps 5
run --N 16384 --tile 512 --steps 5
run --N 12288 --tile 512 --steps 5
run --N 8192 --tile 512 --steps 8
run --N 6144 --tile 512 --steps 8
run --N 4096 --tile 512 --steps 8
run --N 65536 --tile 1024 --dtype f32
run --N 131072 --tile 1024 --dtype f32 --steps 2
cat $out
