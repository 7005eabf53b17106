#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
out=gpurun_out/r03_g.txt
: > $out
timeout -k 10 600 python -m pytest tests/test_gpu_worker.py -m gpu -x -v -p no:cacheprovider --timeout 300 > gpurun_out/r03_g_pytest.log 2>&1
echo "pytest rc=$?" >> $out
tail -3 gpurun_out/r03_g_pytest.log >> $out
timeout -k 10 300 python scripts/worker_profile.py 8192 512 > gpurun_out/r03_worker_profile.txt 2>&1
echo "worker profile rc=$?" >> $out
head -6 gpurun_out/r03_worker_profile.txt >> $out
timeout -k 10 300 python scripts/worker_profile.py 16384 512 > gpurun_out/r03_worker_profile_16k.txt 2>&1
head -6 gpurun_out/r03_worker_profile_16k.txt >> $out
cat $out
