#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
timeout -k 10 200 python -m pytest "tests/test_gpu_worker.py::test_wave_level_execution_reports_the_per_task_statuses" -x -v -p no:cacheprovider -o faulthandler_timeout=45 --timeout 120 > gpurun_out/r03_e_pytest.log 2>&1
echo "rc=$?"
tail -60 gpurun_out/r03_e_pytest.log | cut -c1-220
