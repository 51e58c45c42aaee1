#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters (separate passes, no tracing domains besides kernel-trace).
set -u
W=${1:-cistgcn8_b16_t50_v22}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$W
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/$CNT" -- python3 "$GRAFT_REPO_ROOT/tools/prof_domain_kernel.py" "$W" 10 > "$OUT/$CNT.log" 2>&1
  echo "$CNT rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = {}
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(out + "/" + cnt + "/**/*counter_collection.csv", recursive=True)
    tot, n = 0.0, 0
    for f in files:
        for r in csv.DictReader(open(f)):
            if "cg_stgcn_domain_fwd" in r.get("Kernel_Name", "") and r.get("Counter_Name") == cnt:
                tot += float(r["Counter_Value"]); n += 1
    res[cnt] = {"sum": tot, "dispatches": n}
    print(cnt, "files", len(files), "dispatches", n, "sum", tot)
json.dump(res, open(out + "/summary.json", "w"))
PY
# drop the bulky per-dispatch files, keep the summary
find "$OUT" -name "*.csv" -size +5M -delete
