#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel stats.  Logs under gpurun_out/.
# A step that is killed by its timeout stops the session (no further GPU work after a hang).
set -u
mkdir -p gpurun_out
run() {  # name, timeout, command...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/session.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a gpurun_out/session.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/session.log; exit 1; fi
  return 0
}
: > gpurun_out/session.log
run build 300 python -c "import __graft_entry__ as g; g.build()"
run pytest_gpu 900 python -m pytest tests -m gpu -q --timeout 600
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3
run smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
tail -2 gpurun_out/smoke.log
run bench_default 600 python bench.py --steps 50 --warmup 10
tail -c 300 gpurun_out/bench_default.log
run profile 900 bash tools/gpu_profile.sh cistgcn8_b16_t50_v22
grep -A 14 '"Name"' gpurun_out/profile.log | cut -c1-150
run profile64 900 bash tools/gpu_profile.sh cistgcn64_b256_t50_v22
grep -A 14 '"Name"' gpurun_out/profile64.log | cut -c1-150
