cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in 2048x512 4096x512 4096x1024 8192x1024; do
for f in 0 1; do
echo "== $cfg FLOW=$f"; CHOLMI_FLOW=$f PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $cfg 2>&1 | grep "rep=2"
done; done
