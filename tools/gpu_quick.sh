#!/bin/bash
# One GPU-box call: build, parity tests, smoke, both bench lines, kernel profiles.
# A step that times out or is killed ends the call: nothing else is started on the card after it.
set -u
mkdir -p gpurun_out
step() {  # step <seconds> <log> <cmd...>
  local secs=$1 log=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "$(basename "$log") rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out; stopping"; tail -5 "$log"; exit $rc; fi
  return $rc
}
step 600 gpurun_out/build.log python -c "import __graft_entry__ as g; g.build()" || exit 1
step 900 gpurun_out/pytest_gpu.log python -m pytest tests -m gpu -q --timeout 600
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
grep -E "^E  " gpurun_out/pytest_gpu.log | cut -c1-250 | head -10
step 300 gpurun_out/smoke.log python -c "import __graft_entry__ as g; g.smoke()"; tail -1 gpurun_out/smoke.log
step 400 gpurun_out/bench_default.log python bench.py --steps 50 --warmup 10
grep -E "metric|NaN|Error" gpurun_out/bench_default.log | cut -c1-1500
step 300 gpurun_out/bench_c64.log python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload cistgcn64_b256_t50_v22
grep -E "metric|NaN|Error" gpurun_out/bench_c64.log | cut -c1-300
step 600 gpurun_out/profile.log bash tools/gpu_profile.sh cistgcn8_b16_t50_v22
step 600 gpurun_out/profile64.log bash tools/gpu_profile.sh cistgcn64_b256_t50_v22
exit 0
