#!/bin/bash
# Round-2 evidence run on one MI355X box: GPU tests, the default bench line, rocprofv3 kernel stats of the bench command for
# both workloads, PMC passes (fused stage, tail, whole step by kernel).  Everything lands under gpurun_out/; the summaries
# are copied into profiles/ afterwards (tools/collect_profiles_r02.py).
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m cistgcn_amd.build > gpurun_out/r2_build.log 2>&1 || exit 1
if [ "${1:-all}" != "profiles" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_pytest_gpu.log
  tail -3 gpurun_out/r2_pytest_gpu.log
  timeout -k 10 600 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; head -c 300 gpurun_out/r02_bench.json; echo
fi
cd /tmp
for W in cistgcn64_b256_t50_v22 cistgcn8_b16_t50_v22; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02_$W
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload $W --no-cpu-baseline --no-secondary --steps 40 > "$OUT/bench.log" 2>&1
  echo "profile $W rc=$?"
  find "$OUT" -name "*kernel_trace.csv" -size +5M -delete
done
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc_kernels.sh step_c64 cg_ tools/prof_step.py cistgcn64_b256_t50_v22 2 > gpurun_out/r2_pmc_step.log 2>&1; tail -3 gpurun_out/r2_pmc_step.log
bash tools/gpu_pmc_kernels.sh domain cg_stgcn_domain tools/prof_domain_sq.py 2 > gpurun_out/r2_pmc_domain.log 2>&1; tail -3 gpurun_out/r2_pmc_domain.log
bash tools/gpu_pmc_kernels.sh tail cg_tail tools/prof_tail.py > gpurun_out/r2_pmc_tail.log 2>&1; tail -3 gpurun_out/r2_pmc_tail.log
bash tools/gpu_pmc_kernels.sh adj cg_adj tools/prof_adj.py > gpurun_out/r2_pmc_adj.log 2>&1; tail -3 gpurun_out/r2_pmc_adj.log
