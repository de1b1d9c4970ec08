#!/bin/bash
# round 2, GPU call 35: the launcher paths of bench.py on a real GPU (torch.distributed.run with one rank = RCCL init; --gpus 2 on a one-GPU box must refuse)
export TMPDIR=/tmp
out=$PWD/gpurun_out; mkdir -p $out
{
echo "== torch.distributed.run, 1 rank"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu 2>&1 | tail -2 | cut -c1-400
echo "== python bench.py --gpus 2 on this box (expects a refusal)"
timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu 2>&1 | tail -3 | cut -c1-300; echo "rc=$?"
} 2>&1 | tee $out/c35_launchers.txt
