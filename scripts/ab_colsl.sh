cd ${GRAFT_REPO_ROOT:-/root/repo}
CHOLMI_COL_SLICES=1 timeout -k 10 400 python -m pytest tests/test_gpu_full.py -x -q -k "variants or full_potrf or config2 or flow" > gpurun_out/r04_cs_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04_cs_pytest.log
for cfg in 4096x512 8192x512 16384x512 32768x512 8192x1024 16384x1024; do
for c in 0 1; do
echo -n "$cfg COL_SLICES=$c  "; CHOLMI_COL_SLICES=$c python scripts/probe_perf.py $cfg 2>/dev/null | grep "rep=2" | cut -c1-78
done; done
