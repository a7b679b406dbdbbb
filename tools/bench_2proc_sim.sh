#!/bin/bash
# Exercise bench.py's multi-process path on ONE GPU box: two ranks sharing GPU 0 would
# fight over RCCL device uniqueness, so this only checks argument / rank plumbing with
# WORLD_SIZE=1 under torchrun (the real 2/4/8-GPU runs are the driver's).
export CARL_AMD_NO_BUILD=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 500 --warmup 50 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
