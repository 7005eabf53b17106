#!/bin/bash
# round 3: rest of the GPU suite, the worker path per task, one rank of each grid alone (null transport)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
out=gpurun_out/r03_d.txt
: > $out
timeout -k 10 900 python -m pytest tests/test_gpu_worker.py tests/test_gpu_tileops.py tests/test_gpu_distributed.py -m gpu -x -v -p no:cacheprovider --timeout 400 > gpurun_out/r03_d_pytest.log 2>&1
echo "pytest rc=$?" >> $out
tail -4 gpurun_out/r03_d_pytest.log >> $out
timeout -k 10 300 python scripts/worker_profile.py 8192 512 > gpurun_out/r03_worker_profile.txt 2>&1
echo "worker profile rc=$?" >> $out
head -4 gpurun_out/r03_worker_profile.txt >> $out
timeout -k 10 400 python scripts/dist_issue_time.py 65536 1024 > gpurun_out/r03_rank_alone_N65536.txt 2>&1
echo "rank-alone rc=$?" >> $out
cat gpurun_out/r03_rank_alone_N65536.txt >> $out
cat $out
