#!/bin/bash
# Counter evidence for the shipped kernels on the bench command itself (run on the GPU box):
#   bash scripts/pmc_bench.sh <tag> <N> <tile> [dtype]
# One kernel-trace pass (durations) + four --pmc passes (SQ issue/wait, SQ instruction mix + LDS +
# L2 hit, FETCH_SIZE, WRITE_SIZE -- FETCH and WRITE cannot share a pass, MI355X_MICROARCH.md
# "rocprofv3 PMC slots"), each over ONE factorisation; the program comes straight after `--`.
# Summary -> gpurun_out/pmc_<tag>_summary.txt (copy to profiles/).
set -u
tag=$1; N=$2; tile=$3; dtype=${4:-f64}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmcb_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--N $N --tile $tile --dtype $dtype --steps 1 --warmup 0 --no-cpu-baseline --no-worker-path --no-live-traffic --no-check"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" $args > "$out/trace.log" 2>&1
p1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
p2="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum"
i=0
for ctrs in "$p1" "$p2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 900 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc$i" -- python3 "$root/bench.py" $args > "$out/pmc$i.log" 2>&1
  echo "pass $i done: $(tail -c 300 "$out/pmc$i.log" | tr '\n' ' ' | cut -c1-200)"
done
cd "$root"
python3 scripts/pmc_summary.py "$out" "$tag" "$N" "$tile" "$dtype" | tee gpurun_out/pmc_${tag}_summary.txt
