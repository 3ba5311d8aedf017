#!/bin/bash
# Evaluation batch and training step of the reference's batching (N = 20,778, 10 slots x <= 20 steps, d = 256) against the number of lanes:
# where the small-batch forms (persistent scan, device-sized grids) hand over to the throughput forms.  usage: [D=128] bash scripts/lane_sweep.sh [LOW=gru]
D=${D:-256}
for B in 32 64 128 256 512 1024 2048 4096; do
  env D=$D $1 python3 scripts/cfg2_bench.py $B 100 2>&1 | grep "ms per batch" | head -1
  env N=20778 S=10 L=20 D=$D $1 python3 scripts/train_bench.py $B 20 ragged 2>&1 | tail -1 | cut -c1-75
done
