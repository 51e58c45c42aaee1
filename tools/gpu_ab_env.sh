#!/bin/bash
# A/B of plan switches (environment variables) on one box, two rounds to see the noise.
# usage: gpu_ab_env.sh "VAR=a VAR2=b" "VAR=c" ...
set -u
mkdir -p gpurun_out
for round in 1 2; do
for cfg in "$@"; do
  a=$(env $cfg timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*') || exit 1
  b=$(env $cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --workload cistgcn64_b256_t50_v22 2>&1 | grep -o '"ms_per_step": [0-9.]*') || exit 1
  echo "$cfg: $a | $b"
done
done
