#!/bin/bash
# rocprofv3 over `python3 <prog> <args>`: kernel stats, then SQ / LDS counters and HBM traffic (FETCH_SIZE / WRITE_SIZE) of the
# kernels whose name contains <pattern>, each counter group in its own pass (kernel-trace only, as gpurun requires).
# Usage: gpu_pmc_kernels.sh <tag> <pattern> <prog> [args...]   -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; PAT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$OUT/stats.log" 2>&1 || tail -3 "$OUT/stats.log"
i=0
for CNT in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$OUT/p$i.log" 2>&1 || { echo "pass $i ($CNT) failed"; tail -3 "$OUT/p$i.log"; }
done
python3 - "$OUT" "$PAT" > "$OUT/summary.txt" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
print("# kernel stats (rocprofv3 --kernel-trace --stats), kernels matching '%s'" % pat)
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            print("%-70s calls %5s  avg %10.1f us  total %10.1f us  %5s%%" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if pat not in k: continue
        name = k.split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
print("# PMC counters, average per launch (FETCH_SIZE is reported at half the bytes of wide streaming reads on gfx950: doubled in DESIGN.md)")
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        print("   %-30s %18.0f  (%d launches)" % (c, acc[name][c] / max(1, cnt[(name, c)]), cnt[(name, c)]))
PY
cat "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +2M -delete
