#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
timeout -k 10 500 python scripts/rehearse_stress.py 16384 512 2>&1 | grep grid | tee gpurun_out/r03_rehearse_stress.txt
timeout -k 10 500 python scripts/rehearse_stress.py 32768 1024 2>&1 | grep grid | tee -a gpurun_out/r03_rehearse_stress.txt
