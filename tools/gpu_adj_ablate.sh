#!/bin/bash
# Ablation of the Map2Adj backward kernels (CG_ADJ_DBG mask: 1 weight-gradient product, 2 dh/do product + epilogue, 8 tile prefetch,
# 16 dQ cells): kernel averages from rocprofv3 --kernel-trace --stats for each mask.
cd /tmp && export TMPDIR=/tmp
for M in ${MASKS:-0 1 2 3 8 11 16 27}; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/adj_abl_$M
  rm -rf $OUT; mkdir -p $OUT
  CISTGCN_ABLATION=1 CG_ADJ_DBG=$M timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_adj.py > $OUT/log.txt 2>&1
  python3 - $OUT $M <<'PY'
import csv, glob, sys
out, m = sys.argv[1], sys.argv[2]
r = {}
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cg_adj" in row["Name"]:
            r[row["Name"].split("(")[0].replace("cg_adj_", "").replace("_kernel", "")] = float(row["AverageNs"]) / 1e3
print("mask %2s  " % m + "  ".join("%s %.0f us" % (k, v) for k, v in sorted(r.items())))
PY
done
