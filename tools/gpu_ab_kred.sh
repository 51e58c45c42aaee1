#!/bin/bash
# A/B of the K-reduction planner thresholds (environment only, no rebuild)
python -m cistgcn_amd.build >/dev/null 2>&1 || exit 1
for cfg in "65536 64" "32768 64" "32768 128" "16384 128" "65536 64"; do
  set -- $cfg
  CISTGCN_KRED_MIN_K=$1 CISTGCN_KRED_MAX=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-eval --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('min_k=$1 max=$2 headline %.2f ms  secondary %.3f ms' % (d['ms_per_step'], d['secondary']['ms_per_step']))"
done
