cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
for f in 0 0.1 0.2 0.3 0.5; do
echo "== $cfg FLOW_FACTOR=$f"; CHOLMI_FLOW_FACTOR=$f PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $cfg 2>&1 | grep "rep=2"
done; done
