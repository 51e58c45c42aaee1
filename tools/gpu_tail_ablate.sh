#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for M in ${MASKS:-0 2 1}; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/tail_abl_$M
  rm -rf $OUT; mkdir -p $OUT
  CG_TAIL_DBG=$M timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_tail.py > $OUT/log.txt 2>&1
  python3 - $OUT $M <<'PY'
import csv, glob, sys
out, m = sys.argv[1], sys.argv[2]
r = {}
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cg_tail" in row["Name"]:
            r[row["Name"].split("(")[0].replace("cg_tail_", "").replace("_kernel", "")] = float(row["AverageNs"]) / 1e3
print("mask %2s  " % m + "  ".join("%s %.0f" % (k, v) for k, v in sorted(r.items())))
PY
done
