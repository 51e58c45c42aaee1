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
# bench.py family -> kernel name prefixes in the PMC summary
FAMILY_KERNELS = {
    CONTRACT_FAMILY: ("cg_contract",),
    "row kernels fwd (cg_norm_act_fwd)": ("cg_norm_act_fwd", "cg_chan_stats"),
    "row kernels bwd (cg_norm_act_bwd reduce + apply)": ("cg_norm_act_bwd",),
    "fused ST-GCN stage fwd (cg_stgcn_domain_fwd)": ("cg_stgcn_domain_fwd",),
    "fused ST-GCN stage bwd (cg_stgcn_domain_bwd)": ("cg_stgcn_domain_bwd", "cg_dom_fold"),
    "DSTD_GC tail phases (cg_dstd_tail_fwd/bwd)": ("cg_tail_",),
    "Map2Adj tail phases (cg_map2adj_tail_fwd/bwd)": ("cg_adj_",),
    "stacked tower maps (cg_pointwise_maps_fwd/bwd)": ("cg_pwm_",),
    "frame-collapsing convolutions (cg_collapse_rows_fwd/bwd)": ("cg_rows_",),
    "time-extrapolator convolutions (cg_fpn_conv_fwd/bwd)": ("cg_fpn_",),
    "block statistics (cg_dstd_stats_fwd/bwd)": ("cg_dstd_stats",),
}


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
        fams = json.load(open(bench)).get("roofline", {}).get("per_family", {}) if os.path.exists(bench) else {}
        steps = 2     # tools/prof_step.py <workload> 2
        rec = {"_steps": steps,
               "_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/prof_step.py (two eager training "
                        "steps); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies a wide coalesced read at half its bytes), "
                        "counters in KB; summed over the kernels of a family and divided by the family's C-ABI calls, the unit "
                        "bench.py's per-call algorithmic bytes use"}
        for fam, prefixes in FAMILY_KERNELS.items():
            calls = fams.get(fam, {}).get("launches")
            total_kb, launches, detail = 0.0, 0, {}
            for k, c in pmc.items():
                if any(k.startswith(p) for p in prefixes) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    fetch, n = c["FETCH_SIZE"]
                    write, _ = c["WRITE_SIZE"]
                    total_kb += n * (2.0 * fetch + write)
                    launches += n
                    detail[k] = {"launches": n, "FETCH_SIZE_KB_avg": fetch, "WRITE_SIZE_KB_avg": write}
            if calls and launches:
                rec[fam] = total_kb * 1024.0 / (steps * calls)
                rec["_kernels " + fam] = detail
        json.dump({HEADLINE: rec}, open(os.path.join(DST, "r02_traffic.json"), "w"), indent=1)
        print("wrote r02_traffic.json:", {k[:24]: round(v / 1e6, 1) for k, v in rec.items() if isinstance(v, float)})


if __name__ == "__main__":
    main()
