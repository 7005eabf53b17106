#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
( TRACE=1 timeout -k 5 260 python scripts/exp/gloo_grid_hang.py 3 1 > gpurun_out/r03_hang31.log 2>&1; echo "3x1 rc=$?" >> gpurun_out/r03_hang31.log )
tail -40 gpurun_out/r03_hang31.log
