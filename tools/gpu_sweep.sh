#!/bin/bash
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for cfg in "64 256 50 22" "8 16 50 22"; do
  echo "== $cfg"
  timeout -k 10 120 python tools/bench_domain.py $cfg 2>&1 | grep "dom"
  for gt in 1 2 4; do for per in 1 4 16; do
    CG_DOM_GT=$gt CG_DOM_PER=$per timeout -k 10 120 python tools/bench_domain.py $cfg 2>&1 | grep "dom"
  done; done
done
