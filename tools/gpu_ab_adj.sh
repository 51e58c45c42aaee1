#!/bin/bash
# A/B of the fused Map2Adj tail (CISTGCN_FUSED_ADJ) on one box: operator check, model parity, bench both ways.
set -o pipefail
mkdir -p gpurun_out
python -m cistgcn_amd.build > gpurun_out/ab_adj_build.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "map2adj or golden or strict or branches or full_size or kernel_of_a_step" > gpurun_out/ab_adj_pytest.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/ab_adj_pytest.log
for v in 1 0 1 0; do
  CISTGCN_FUSED_ADJ=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-eval --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused_adj=$v headline %.2f ms  secondary %.3f ms' % (d['ms_per_step'], d['secondary']['ms_per_step']))"
done
timeout -k 10 200 python tools/probe_calls.py > gpurun_out/ab_adj_probe.log 2>&1; tail -40 gpurun_out/ab_adj_probe.log
