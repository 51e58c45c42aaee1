#!/bin/bash
# Short GPU session: parity tests + smoke + quick bench + 2-rank rehearsal of the N>1 path (gloo, both ranks on the one GPU).
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
grep -E "^E  " gpurun_out/pytest_gpu.log | cut -c1-250 | head -10
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_quick.log 2>&1; echo "bench rc=$?"; tail -c 700 gpurun_out/bench_quick.log
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo > gpurun_out/bench_dp2_gloo.log 2>&1; echo "dp2 rc=$?"; tail -c 700 gpurun_out/bench_dp2_gloo.log
