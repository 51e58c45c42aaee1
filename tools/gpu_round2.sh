#!/bin/bash
# One GPU-box session of round 2.  Logs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m cistgcn_amd.build > gpurun_out/r2_build.log 2>&1 || exit 1
timeout -k 10 300 python tools/bench_domain_bwd.py > gpurun_out/r2_bench_domain_bwd.log 2>&1; cat gpurun_out/r2_bench_domain_bwd.log | tail -14
timeout -k 10 200 python tools/trace_aten.py > gpurun_out/r2_trace_aten.log 2>&1; tail -25 gpurun_out/r2_trace_aten.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_pytest_gpu.log
tail -5 gpurun_out/r2_pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; tail -c 3000 gpurun_out/r2_bench.json; tail -5 gpurun_out/r2_bench.err
