#!/bin/bash
# Rehearsal of the N>1 bench flow on a 1-GPU box: 2 ranks share the card, exchange over gloo (host staged).
# Not a measurement: the measured configuration is one rank per GPU over RCCL (the driver runs it).
set -e
# (RCCL refuses two ranks on one device, so the library's own transport cannot be rehearsed this way: see bench.py --self-partition)
export HFX_BENCH_TRANSPORT=gloo
python bench.py --gpus 2 --steps 3 --warmup 1 --cells ${1:-16}
