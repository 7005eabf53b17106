#!/bin/bash
# round 4: race screens with the new defaults (near column, latency-form column update), then with the streaming row slabs and with 8192 / 12288 (late flow waves)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 500 python scripts/stress_flow.py > gpurun_out/r04_stress_flow2.txt 2>&1; echo "default rc=$?"; tail -3 gpurun_out/r04_stress_flow2.txt
CHOLMI_FLOW_ROWS=2 timeout -k 10 500 python scripts/stress_flow.py > gpurun_out/r04_stress_flow2_rows2.txt 2>&1; echo "rows2 rc=$?"; tail -3 gpurun_out/r04_stress_flow2_rows2.txt
timeout -k 10 400 python scripts/stress.py > gpurun_out/r04_stress_std2.txt 2>&1; echo "std rc=$?"; tail -4 gpurun_out/r04_stress_std2.txt
