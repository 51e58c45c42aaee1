#!/bin/bash
# Every bench workload once (short), to catch shape-specific failures: prints ms/step and loss.
set -u
for w in cistgcn8_b16_t50_v22 cistgcn64_b256_t50_v22 cistgcn32_b256_t50_v25 cistgcn8_b16_t10_v22 cistgcn64_b256_t10_v22 cistgcn32_b256_t10_v18; do
  out=$(timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --workload $w 2>&1 | grep '^{') || { echo "$w FAILED"; exit 1; }
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('%-26s %8.3f ms  %9.1f seq/s  eval %9.1f seq/s  loss %.4f' % (d['config']['workload'], d['ms_per_step'], d['value'], d['eval_forward']['value'], d['config']['loss']))" "$out"
done
