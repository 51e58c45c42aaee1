#!/bin/bash
# rocprofv3 kernel stats of the bench command for one workload -> gpurun_out/kstats_<workload>.csv (top of it printed)
W=${1:-cistgcn64_b256_t50_v22}
python -m cistgcn_amd.build >/dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_$W
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload $W --no-cpu-baseline --no-secondary --no-roofline --no-eval --steps 40 > "$OUT/bench.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:32]:
        print("%6.2f%%  calls %6s  avg %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"].split("(")[0][:60]))
PY
