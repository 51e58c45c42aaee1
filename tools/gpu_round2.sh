#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m cistgcn_amd.build > gpurun_out/r2_build.log 2>&1 || exit 1
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_pytest_gpu.log
tail -4 gpurun_out/r2_pytest_gpu.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; head -c 500 gpurun_out/r2_bench.json; tail -2 gpurun_out/r2_bench.err
timeout -k 10 200 python tools/probe_calls.py cistgcn64_b256_t50_v22 45 > gpurun_out/r2_probe_c64.log 2>&1
