#!/bin/bash
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 600 -k "graph or wide" > gpurun_out/pytest_gpu_quick.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/pytest_gpu_quick.log | tail -2
grep -E "^E  " gpurun_out/pytest_gpu_quick.log | cut -c1-250 | head -10
for extra in "" "--no-branches"; do
  timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline $extra 2>&1 | grep -E "metric|NaN|Error" | cut -c1-230
done
for extra in "" "--no-branches"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --workload cistgcn64_b256_t50_v22 $extra 2>&1 | grep -E "metric|NaN|Error" | cut -c1-230
done
