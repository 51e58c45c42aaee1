#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
echo "--- dp2 gloo graph"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 30 --warmup 5 --backend gloo 2>&1 | grep -E "metric|NaN|Warn" | cut -c1-300
echo "--- single, default"
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | grep -E "metric|NaN|Warn" | cut -c1-250
