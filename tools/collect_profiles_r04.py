"""Copies the summaries of tools/gpu_final_r04.sh from gpurun_out/ (scratch) into profiles/ (tracked) under round-3 names and
derives profiles/r04_traffic.json: HBM bytes per training step of every kernel family of bench.py (rocprofv3 --pmc FETCH_SIZE and
--pmc WRITE_SIZE in separate passes over two eager steps of the headline workload; FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for wide streaming reads on gfx950).

    python tools/collect_profiles_r04.py
"""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

SRC = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles")
STEPS = 2          # tools/prof_step.py <workload> 2
BLOCK_REPS = 3     # tools/prof_block.py 64,256,50,22 3


def copy(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(DST, dst))
        print("copied", dst)
    else:
        print("missing", src)


def parse_pmc(path):
    """summary.txt of tools/gpu_pmc_kernels.sh -> {kernel: {counter: (average per launch, launches)}}"""
    out, cur, in_pmc = {}, None, False
    for line in open(path):
        if line.startswith("# PMC"):
            in_pmc = True
            continue
        if not in_pmc:
            continue
        m = re.match(r"\s+(\S+)\s+(\d+)\s+\((\d+) launches\)", line)
        if m and cur is not None:
            out[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
        elif line.strip():
            cur = line.strip()
            out[cur] = {}
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    copy(os.path.join(SRC, "r04_bench.json"), "r04_bench_default.json")
    for w in (bench.HEADLINE, bench.SECONDARY, bench.AMASS25):
        found = sorted(glob.glob(os.path.join(SRC, "prof_r04_" + w, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if found:                                  # gpurun merges into gpurun_out/: earlier runs' files (other pids) stay around
            copy(found[-1], "r04_%s_kernel_stats.csv" % w)
    copy(os.path.join(SRC, "pmc_r04_step_c64", "summary.txt"), "r04_step_c64_pmc.txt")
    copy(os.path.join(SRC, "pmc_r04_block", "summary.txt"), "r04_block_c64_pmc.txt")
    step = os.path.join(SRC, "pmc_r04_step_c64", "summary.txt")
    if not os.path.exists(step):
        return
    pmc = parse_pmc(step)
    rec = {"_steps": STEPS,
           "_note": "bytes per TRAINING STEP and kernel family: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                    "tools/prof_step.py (two eager steps); counters in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies a "
                    "wide coalesced read at half its bytes); summed over the kernels of a family (bench.FAMILY_KERNELS prefixes), divided "
                    "by the steps"}
    total = 0.0
    for fam, prefixes in bench.FAMILY_KERNELS.items():
        kb, detail = 0.0, {}
        for k, c in pmc.items():
            if any(k.startswith(p) for p in prefixes) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                fetch, n = c["FETCH_SIZE"]
                write, _ = c["WRITE_SIZE"]
                kb += n * (2.0 * fetch + write)
                detail[k] = {"launches": n, "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": write}
        if detail:
            rec[fam] = kb * 1024.0 / STEPS
            rec["_kernels " + fam] = detail
            total += rec[fam]
    other = 0.0
    for k, c in pmc.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c and not any(k.startswith(p) for pre in bench.FAMILY_KERNELS.values() for p in pre):
            other += c["FETCH_SIZE"][1] * (2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0])
    rec["_other_kernels"] = other * 1024.0 / STEPS
    rec["_step_total"] = total + rec["_other_kernels"]
    # one DSTD_GC 64 -> 64 block, forward + backward (tools/prof_block.py, BLOCK_REPS invocations): every kernel of it
    blk = os.path.join(SRC, "pmc_r04_block", "summary.txt")
    if os.path.exists(blk):
        kb, detail = 0.0, {}
        for k, c in parse_pmc(blk).items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                kb += c["FETCH_SIZE"][1] * (2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0])
                detail[k] = {"launches": c["FETCH_SIZE"][1], "FETCH_SIZE_KB_avg": c["FETCH_SIZE"][0], "WRITE_SIZE_KB_avg": c["WRITE_SIZE"][0]}
        rec["_block"] = {"what": "HBM bytes of ONE DSTD_GC 64->64 invocation (B=256, T=50, V=22, train, dropout 0.1), forward + backward, all kernels: "
                                 "sum of launches x (2 FETCH_SIZE + WRITE_SIZE) over tools/prof_block.py / %d invocations" % BLOCK_REPS,
                         "bytes": kb * 1024.0 / BLOCK_REPS, "kernels": detail}
    json.dump({bench.HEADLINE: rec}, open(os.path.join(DST, "r04_traffic.json"), "w"), indent=1)
    print("wrote r04_traffic.json: step total %.1f GB; " % (rec["_step_total"] / 1e9),
          {k[:28]: round(v / 1e9, 2) for k, v in rec.items() if isinstance(v, float) and not k.startswith("_")})


if __name__ == "__main__":
    main()
