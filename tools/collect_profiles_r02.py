"""Copies the summaries of tools/gpu_final_r02.sh from gpurun_out/ (scratch) into profiles/ (tracked) under round-2 names and
derives profiles/r02_traffic.json (HBM bytes per launch of the dominant kernel family) from the PMC passes.

    python tools/collect_profiles_r02.py
"""
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles")
HEADLINE = "cistgcn64_b256_t50_v22"
CONTRACT_FAMILY = "contraction (cg_contract_many: tiled / streaming / K-reduction kernels)"


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
    copy(os.path.join(SRC, "r02_bench.json"), "r02_bench_default.json")
    for w in (HEADLINE, "cistgcn8_b16_t50_v22"):
        found = sorted(glob.glob(os.path.join(SRC, "prof_r02_" + w, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if found:                                  # gpurun merges into gpurun_out/: earlier runs' files (other pids) stay around
            copy(found[-1], "r02_%s_kernel_stats.csv" % w)
    copy(os.path.join(SRC, "pmc_step_c64", "summary.txt"), "r02_step_c64_pmc.txt")
    copy(os.path.join(SRC, "pmc_domain", "summary.txt"), "r02_stgcn_domain_pmc.txt")
    copy(os.path.join(SRC, "pmc_tail", "summary.txt"), "r02_dstd_tail_pmc.txt")
    copy(os.path.join(SRC, "pmc_adj", "summary.txt"), "r02_map2adj_tail_pmc.txt")
    step = os.path.join(SRC, "pmc_step_c64", "summary.txt")
    if os.path.exists(step):
        pmc = parse_pmc(step)
        bench = os.path.join(SRC, "r02_bench.json")
        calls_per_step = None
        if os.path.exists(bench):
            fam = json.load(open(bench)).get("roofline", {}).get("per_family", {}).get(CONTRACT_FAMILY)
            calls_per_step = fam and fam["launches"]
        total_kb, launches, detail = 0.0, 0, {}
        for k, c in pmc.items():
            if k.startswith("cg_contract") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                fetch, n = c["FETCH_SIZE"]
                write, _ = c["WRITE_SIZE"]
                total_kb += n * (2.0 * fetch + write)
                launches += n
                detail[k] = {"launches": n, "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": write}
        steps = 2     # tools/prof_step.py <workload> 2
        per_call = total_kb * 1024.0 / (steps * calls_per_step) if calls_per_step else None
        rec = {HEADLINE: {CONTRACT_FAMILY: per_call, "_kernels": detail, "_kernel_launches": launches, "_steps": steps,
                          "_calls_per_step": calls_per_step,
                          "_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_step.py (two eager "
                                   "training steps); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies a wide coalesced read at "
                                   "half its bytes), counters in KB; summed over the contraction kernels and divided by the C-ABI calls of "
                                   "the family, the unit bench.py's per-call algorithmic bytes use"}}
        json.dump(rec, open(os.path.join(DST, "r02_traffic.json"), "w"), indent=1)
        print("wrote r02_traffic.json: %.1f MB per call" % ((per_call or 0) / 1e6))


if __name__ == "__main__":
    main()
