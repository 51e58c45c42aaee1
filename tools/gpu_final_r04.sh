#!/bin/bash
# Round-4 evidence runs on one MI355X box (a gpurun call is limited to 20 minutes, so the work is cut in parts):
#   gpu_final_r04.sh tests     GPU test suite -> gpurun_out/r4_pytest_gpu.log
#   gpu_final_r04.sh bench     the default bench line -> gpurun_out/r04_bench.json
#   gpu_final_r04.sh stats     rocprofv3 --kernel-trace --stats of the bench command, headline / secondary / configs[4]-shape workloads
#   gpu_final_r04.sh pmc_step  PMC passes (SQ, LDS, FETCH_SIZE, WRITE_SIZE, each in its own pass) over two eager steps, by kernel
#   gpu_final_r04.sh pmc_block the same over three invocations of ONE DSTD_GC 64 -> 64 block (tools/prof_block.py: the unit of `roofline`)
# Everything lands under gpurun_out/; tools/collect_profiles_r04.py copies the summaries into profiles/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
case "${1:-tests}" in
tests)
  timeout -k 10 1150 python -m pytest tests -m gpu -x -q -s > gpurun_out/r4_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r4_pytest_gpu.log
  tail -3 gpurun_out/r4_pytest_gpu.log ;;
bench)
  timeout -k 10 1100 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; head -c 300 gpurun_out/r04_bench.json; echo ;;
stats)
  cd /tmp
  for W in cistgcn64_b256_t50_v22 cistgcn8_b16_t50_v22 cistgcn32_b256_t50_v25; do
    OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04_$W
    rm -rf "$OUT"; mkdir -p "$OUT"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload $W --no-cpu-baseline --no-secondary --no-dp-overhead --steps 40 > "$OUT/bench.log" 2>&1
    echo "profile $W rc=$?"
    find "$OUT" -name "*kernel_trace.csv" -size +5M -delete
  done ;;
pmc_step)
  bash tools/gpu_pmc_kernels.sh r04_step_c64 cg_ tools/prof_step.py cistgcn64_b256_t50_v22 2 > gpurun_out/r4_pmc_step.log 2>&1; tail -3 gpurun_out/r4_pmc_step.log ;;
pmc_block)
  bash tools/gpu_pmc_kernels.sh r04_block cg_ tools/prof_block.py 64,256,50,22 3 > gpurun_out/r4_pmc_block.log 2>&1; tail -3 gpurun_out/r4_pmc_block.log ;;
esac
