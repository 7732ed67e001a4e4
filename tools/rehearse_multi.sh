#!/bin/bash
# Rehearsal of the N>1 bench flow on a 1-GPU box: 2 ranks share the card, exchange over gloo (host staged).
# Not a measurement: the measured configuration is one rank per GPU over RCCL (the driver runs it).
set -e
export HFX_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 3 --warmup 1 --cells ${1:-16}
