cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in 1024x256 2048x256 4096x256 1536x512 3072x512 5120x512 6144x512 2048x1024 4096x1024 6144x1024; do
for f in 0 1.0; do
echo "== $cfg FLOW_FACTOR=$f"; CHOLMI_FLOW_MAX_NBM=8 CHOLMI_FLOW_FACTOR=$f PROBE_QUICK=0 timeout -k 10 120 python scripts/probe_perf.py $cfg 2>&1 | grep "rep=2"
done; done
