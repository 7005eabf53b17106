#!/bin/bash
# bench.py --gpus N rehearsed on ONE GPU: the ranks share cuda:0, tiles move over gloo (RCCL refuses two ranks on one device)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 CHOLMI_DIST_BACKEND=gloo
for n in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --N 8192 --tile 512 --steps 2 --warmup 1 > gpurun_out/r03_bench_gloo_$n.log 2>&1
  echo "n=$n rc=$?"
  grep '"metric"' gpurun_out/r03_bench_gloo_$n.log | cut -c1-900
done
