cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in 32768x1024 65536x1024 32768x512; do
for c in 0 1 0 1; do
echo -n "$cfg PERSIST=$c  "; CHOLMI_PERSIST=$c python scripts/probe_perf.py $cfg 2>/dev/null | grep "rep=2" | cut -c1-120
done; done
CHOLMI_PERSIST=1 python scripts/chk_residuals.py 2>&1 | tail -3
