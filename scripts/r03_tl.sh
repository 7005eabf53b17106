#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/trace_cfg.sh r03_16k 16384x512 > gpurun_out/r03_tl.log 2>&1
bash scripts/trace_cfg.sh r03_32k 32768x512 >> gpurun_out/r03_tl.log 2>&1
head -40 gpurun_out/timeline_r03_16k.txt
