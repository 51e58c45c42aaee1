#!/bin/bash
# SQ / LDS counters of the fused ST-GCN kernels (64->64 channels, B=256): separate PMC passes, kernel-trace only.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$GRAFT_REPO_ROOT/tools/prof_domain_sq.py" 2 > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; }
done
python3 - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "cg_stgcn_domain" not in k: continue
        name = k.split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(name, r["Counter_Name"])] += 1
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        print("   %-24s %16.0f  per launch (%d)" % (c, acc[name][c] / max(1, cnt[(name, c)]), cnt[(name, c)]))
PY
cat "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +2M -delete
