#!/bin/bash
# A/B of prebuilt library variants (variants/*.so) on one box, two rounds to see the noise.
set -u
cp cistgcn_amd/libcistgcn_hip.so /tmp/libcistgcn_hip.so.built
trap 'cp /tmp/libcistgcn_hip.so.built cistgcn_amd/libcistgcn_hip.so' EXIT      # leave the library that was built from the tree
for round in 1 2; do
for v in variants/*.so; do
  cp "$v" cistgcn_amd/libcistgcn_hip.so
  a=$(timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*') || exit 1
  b=$(timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --workload cistgcn64_b256_t50_v22 2>&1 | grep -o '"ms_per_step": [0-9.]*') || exit 1
  echo "$v: $a | $b"
done
done
