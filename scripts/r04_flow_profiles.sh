#!/bin/bash
# round 4: the flow form of the tile POTRF -- in-kernel stamps of the chain and the end-to-end table (GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r04_flow_step_stamps.txt
{
echo "# In-kernel s_memrealtime stamps of the diagonal-block workgroups (chol_debug_stamps), flow form (k_flow_factor / k_flow_rows)"
echo "# against the launch-per-step form (CHOLMI_FLOW=0); scripts/flow_stamps.py.  'step' = start of one block's factorisation to the next one's."
for cfg in 4096x512 2048x512; do
echo "## flow form, $cfg"; python scripts/flow_stamps.py $cfg 3 2>/dev/null
echo "## launch-per-step form (CHOLMI_FLOW=0), $cfg"; CHOLMI_FLOW=0 python scripts/flow_stamps.py $cfg 3 2>/dev/null
echo "## flow form with the streaming row slabs (CHOLMI_FLOW_ROWS=2), $cfg"; CHOLMI_FLOW_ROWS=2 python scripts/flow_stamps.py $cfg 3 2>/dev/null
done
} > $out
tab=gpurun_out/r04_flow_end_to_end.txt
{
echo "# whole factorisation, device time of the third repetition (scripts/probe_perf.py), flow form by its default rule vs CHOLMI_FLOW=0"
for cfg in 1536x512 2048x512 3072x512 4096x512 5120x512 6144x512 7168x512 8192x512 10240x512 12288x512 16384x512 3072x384 6144x384 4096x1024; do
for f in 1 0; do
echo -n "$cfg CHOLMI_FLOW=$f  "; CHOLMI_FLOW=$f python scripts/probe_perf.py $cfg 2>/dev/null | grep "rep=2" | cut -c1-78
done; done
} > $tab
cat $tab
