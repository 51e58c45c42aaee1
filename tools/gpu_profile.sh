#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command (default workload unless $1 names another).
set -u
W=${1:-cistgcn8_b16_t50_v22}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$W
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --workload "$W" > "$OUT/bench.log" 2>&1
echo "rc=$?"
tail -c 400 "$OUT/bench.log"
find "$OUT" -name "*kernel_stats.csv" | head -2
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200
# keep the merged output small: drop the per-dispatch trace, keep the stats
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
