#!/bin/bash
# round 3: the whole GPU suite on the unified walker, then the timing table
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
out=gpurun_out/r03_c.txt
: > $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -v -p no:cacheprovider > gpurun_out/r03_c_pytest.log 2>&1
echo "pytest rc=$?" >> $out
tail -4 gpurun_out/r03_c_pytest.log >> $out
run() { python bench.py --no-cpu-baseline --no-check "$@" 2>>gpurun_out/r03_c_err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d.get('roofline',{}); print(d['config']['N'], d['config']['tile'], d['dtype'], d['value'], 'TF/s', d['ms_per_step'], 'ms; upd', r.get('achieved'), 'probe', r.get('peak_probe'), d['config'].get('schedule_calibration'))" >> $out; }
for cfg in "--N 65536 --tile 1024 --steps 3" "--N 32768 --tile 512 --steps 3" "--N 16384 --tile 512 --steps 5" "--N 8192 --tile 512 --steps 8" "--N 65536 --tile 1024 --dtype f32 --steps 3" "--N 32768 --tile 512 --dtype f32 --steps 3"; do
  run $cfg
  echo "-- pair_start=0:" >> $out; CHOLMI_PAIR_START=0 run $cfg
done
cat $out
