cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in 0 0.01 0.03 0.05 0.07 0.1; do
echo "== FLOW_FACTOR=$f"; CHOLMI_FLOW_FACTOR=$f PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py 16384x512 2>&1 | grep "rep=[12]" | head -2
done
echo "== FLOW=0"; CHOLMI_FLOW=0 PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py 16384x512 2>&1 | grep "rep=[12]" | head -2
