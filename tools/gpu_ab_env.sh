#!/bin/bash
# A/B of one environment switch on one box: gpu_ab_env.sh VAR [bench args...] -> headline / secondary ms per step for VAR=1 and VAR=0
VAR=$1; shift
python -m cistgcn_amd.build >/dev/null 2>&1 || exit 1
for v in 1 0 1 0; do
  env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-eval --steps 40 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v headline %.2f ms  secondary %.3f ms' % (d['ms_per_step'], d['secondary']['ms_per_step']))"
done
