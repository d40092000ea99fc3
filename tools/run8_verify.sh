#!/bin/bash
# `bench.py --gpus 8 --verify` <n> times with eight ranks on ONE GPU (gloo collectives): every rank recomputes every shard and compares it
# with the gathered rows, bit for bit.  usage: tools/run8_verify.sh <n> [extra bench args]   (run from the repository root on a GPU box)
n=$1; shift
mkdir -p gpurun_out/r05b
for i in $(seq 1 $n); do
  SVT_VERIFY_DIAG=1 SVT_SHARE_GPU=1 SVT_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --gpus 8 --steps 3 --warmup 1 --batch 2 --seconds 2 --no-cpu-baseline --no-extra-legs --verify "$@" 2>gpurun_out/r05b/err_$i.txt | grep '^{' | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('run $i verified', r['verified'])"
  grep "differ\|diag" gpurun_out/r05b/err_$i.txt | head -24
done
