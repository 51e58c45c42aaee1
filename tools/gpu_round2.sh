#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m cistgcn_amd.build > gpurun_out/r2_build.log 2>&1 || exit 1
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_pytest_gpu.log
tail -4 gpurun_out/r2_pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; head -c 600 gpurun_out/r2_bench.json; tail -2 gpurun_out/r2_bench.err
CISTGCN_KRED_MAX=128 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-eval --no-roofline > gpurun_out/r2_bench_kred128.json 2>/dev/null; head -c 400 gpurun_out/r2_bench_kred128.json
