"""Fills the @PLACEHOLDERS@ of DESIGN.md section 7 from profiles/r02_bench_default.json (run after tools/collect_profiles_r02.py)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_default.json")))
sec, ev, rf = d["secondary"], d["eval_forward"], d["roofline"]
cpu_h, cpu_s = d["cpu_baseline"]["value"], sec["cpu_baseline"]["value"]
fams = sorted(rf["per_family"].items(), key=lambda kv: -kv[1]["us"])
tot = sum(f["us"] for _, f in fams)
rows = ["| family (C-ABI entry points) | calls | time | share | algorithmic GB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
for k, f in fams:
    gb = f.get("GBps")
    rows.append("| %s | %d | %.2f ms | %.0f %% | %s | %s |" % (k, f["launches"], f["us"] / 1e3, 100 * f["us"] / tot,
                                                              ("%.0f" % gb) if gb else "—", ("%.1f %%" % (100 * f["frac"])) if gb else "—"))
rep = {"@H_MS@": "%.1f" % d["ms_per_step"], "@H_SEQ@": "{:,.0f}".format(d["value"]).replace(",", " "),
       "@E_MS@": "%.2f" % ev["ms_per_batch"], "@E_SEQ@": "{:,.0f}".format(ev["value"]).replace(",", " "),
       "@CPU_H@": "%.1f" % cpu_h, "@X_H@": "%.0f" % (d["value"] / cpu_h),
       "@S_MS@": "%.2f" % sec["ms_per_step"], "@S_SEQ@": "{:,.0f}".format(sec["value"]).replace(",", " "),
       "@SE_MS@": "%.2f" % sec["eval_forward"]["ms_per_batch"], "@SE_SEQ@": "{:,.0f}".format(sec["eval_forward"]["value"]).replace(",", " "),
       "@CPU_S@": "%.1f" % cpu_s, "@X_S@": "%.0f" % (sec["value"] / cpu_s),
       "@DOM@": rf["kernel"], "@DOM_FRAC@": "%.1f %%" % (100 * rf["frac"]),
       "@BLK_FRAC@": "%.1f %%" % (100 * d["block_roofline"]["frac"]), "@STEP_FRAC@": "%.1f %%" % (100 * d["step_roofline"]["frac"]),
       "@FAMILIES@": "\n".join(rows)}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for k, v in rep.items():
    s = s.replace(k, v)
open(p, "w").write(s)
print("filled", [k for k in rep])
