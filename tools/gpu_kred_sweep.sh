#!/bin/bash
# K-reduction tuning sweep (GPU box): microbenchmark + step time under a few plan settings.
set -u
mkdir -p gpurun_out
for cfg in "1 64 256" "1 64 128" "0 64 256"; do
  set -- $cfg
  export CISTGCN_KRED=$1 CISTGCN_KRED_MAX=$2 CISTGCN_KRED_BLOCKS=$3
  tag="kred$1_max$2_blk$3"
  timeout -k 10 200 python tools/bench_contract.py > gpurun_out/bc64_$tag.log 2>&1 || exit 1
  timeout -k 10 200 python tools/bench_contract.py 8 16 50 22 > gpurun_out/bc8_$tag.log 2>&1 || exit 1
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/b8_$tag.log 2>&1 || exit 1
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --workload cistgcn64_b256_t50_v22 > gpurun_out/b64_$tag.log 2>&1 || exit 1
  echo "$tag: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b8_$tag.log) | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b64_$tag.log)"
  grep -E "dW" gpurun_out/bc64_$tag.log gpurun_out/bc8_$tag.log | grep -v rows | awk '{print $1,$2,$3,$(NF-3),$(NF-2),$(NF-4)}'
done
