#!/usr/bin/env python3
"""One eager training step with every C-ABI call bracketed by HIP events: the calls aggregated by (entry point, algorithmic
bytes) - i.e. by call site and shape - and the timeline in launch order.  Which passes cost what.  GPU box only."""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cistgcn_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cistgcn64_b256_t50_v22")
ap.add_argument("--top", type=int, default=60)
ap.add_argument("--timeline", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
from cistgcn_amd.models import CISTGCN_0
from cistgcn_amd.runtime import EagerStep
C, B, T, V = bench.WORKLOADS[a.workload]
torch.manual_seed(0)
model = CISTGCN_0(*bench.make_cfg(C, T, V, 0.1)).to(dev).train()
ops.manual_seed(1234, dev)
x, tgt = [t.to(dev) for t in bench.synth(B, T, V, 0)]
step = EagerStep(model, x, tgt)
for _ in range(3):
    step.replay()
torch.cuda.synchronize()
rows = []
orig = _lib.call


def call(name, *args):
    fam, nbytes = bench._call_bytes(name, args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(name, *args)
    e1.record()
    tag = name
    if name == "cg_contract_many":
        tag += " " + " | ".join("G%d M%d N%d K%d sk%d" % (args[0][i].G, args[0][i].M, args[0][i].N, args[0][i].K, args[0][i].splitk) for i in range(args[1]))
    elif name in ("cg_norm_act_fwd_many", "cg_norm_act_bwd_many"):
        n = args[-2] if name.endswith("bwd_many") else args[1]
        tag += " " + " | ".join("%s s%s bn%d%s%s%s%s" % (tuple(args[0][i].xv.n), tuple(args[0][i].xv.s), args[0][i].bn_mode, " pre" if args[0][i].pre else "",
                                                      " add" if args[0][i].add else "", " drop" if args[0][i].drop_p > 0 else "",
                                                      " ystats" if args[0][i].ystats else "") for i in range(n))
    elif name == "cg_sum_many":
        tag += " n=%d %s" % (args[3], tuple(args[1]._obj.n))
    elif name.startswith("cg_dstd_tail") or name.startswith("cg_map2adj_tail"):
        tag += " phase %d" % (args[1] if name.startswith("cg_dstd") else args[2])
    rows.append((tag, nbytes, e0, e1))


_lib.call = call
step.replay()
_lib.call = orig
torch.cuda.synchronize()
tl = [(n, b, e0.elapsed_time(e1) * 1e3) for n, b, e0, e1 in rows]
tot = sum(t for _, _, t in tl)
print("%d calls, %.1f us" % (len(tl), tot))
agg = collections.defaultdict(lambda: [0, 0.0])
for n, b, t in tl:
    agg[(n, b)][0] += 1
    agg[(n, b)][1] += t
cum = 0.0
for (n, b), (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    cum += t
    print("%9.1f us  x%-3d %8.1f us each  %7.1f MB  %6.0f GB/s  cum %5.1f%%  %s" % (t, k, t / k, b / 1e6, b / max(t / k, 1e-9) / 1e3, 100 * cum / tot, n[:400]))
if a.timeline:
    c = 0.0
    for i, (n, b, t) in enumerate(tl):
        c += t
        print("%4d %9.1f  %8.1f us %7.1f MB  %s" % (i, c, t, b / 1e6, n[:400]))
