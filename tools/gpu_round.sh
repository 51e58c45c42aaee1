#!/bin/bash
# Full GPU-box session: parity tests, smoke, bench (with CPU baseline), rocprof kernel stats for both workloads, PMC traffic.
set -u
mkdir -p gpurun_out
step() { echo "=== $1 $(date +%H:%M:%S)"; }
step build; python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
step pytest; timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
grep -E "^E  " gpurun_out/pytest_gpu.log | cut -c1-250 | head -10
step smoke; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
step bench; timeout -k 10 400 python bench.py --steps 50 --warmup 10 > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; grep -E "metric|NaN|Error" gpurun_out/bench_default.log | cut -c1-400
step bench64; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload cistgcn64_b256_t50_v22 > gpurun_out/bench_c64.log 2>&1; grep -E "metric|NaN|Error" gpurun_out/bench_c64.log | cut -c1-300
step profile; bash tools/gpu_profile.sh cistgcn8_b16_t50_v22 > gpurun_out/profile.log 2>&1; echo "profile rc=$?"
step profile64; bash tools/gpu_profile.sh cistgcn64_b256_t50_v22 > gpurun_out/profile64.log 2>&1; echo "profile64 rc=$?"
step pmc; bash tools/gpu_pmc.sh cistgcn8_b16_t50_v22 > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log
step pmc64; bash tools/gpu_pmc.sh cistgcn64_b256_t50_v22 > gpurun_out/pmc64.log 2>&1; tail -3 gpurun_out/pmc64.log
step done
