#!/bin/bash
# One GPU-box session: bench + rocprof kernel stats for both workloads (+ PMC traffic).  Logs under gpurun_out/.
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
run bench_default 600 python bench.py --steps 50 --warmup 10
tail -c 400 gpurun_out/bench_default.log
run profile 900 bash tools/gpu_profile.sh cistgcn8_b16_t50_v22
run profile64 900 bash tools/gpu_profile.sh cistgcn64_b256_t50_v22
run pmc 600 bash tools/gpu_pmc.sh cistgcn8_b16_t50_v22
tail -5 gpurun_out/pmc.log
run pmc64 600 bash tools/gpu_pmc.sh cistgcn64_b256_t50_v22
tail -5 gpurun_out/pmc64.log
