#!/usr/bin/env python3
"""GPU box: HIP-event time of every C-ABI call of one eager training step, grouped by entry point and shape
(contractions: (mode, G, M, N, K) of each problem of the launch).  Usage: probe_calls.py [workload] [top]"""
import collections, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cistgcn_amd import _lib, ops
from cistgcn_amd.models import CISTGCN_0
from cistgcn_amd.runtime import EagerStep

wl = sys.argv[1] if len(sys.argv) > 1 else bench.HEADLINE
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
C, B, T, V = bench.WORKLOADS[wl]
torch.manual_seed(0)
net = CISTGCN_0(*bench.make_cfg(C, T, V, 0.1)).cuda().train()
x, tgt = bench.synth(B, T, V, 0)
step = EagerStep(net, x.cuda(), tgt.cuda())
for _ in range(3):
    step.replay()
torch.cuda.synchronize()
rows = []
orig = _lib.call


def describe(name, args):
    if name == "cg_contract_many":
        arr, n = args[0], args[1]
        return " ".join("[m%d G%d M%d N%d K%d s%d]" % (arr[i].mode, arr[i].G, arr[i].M, arr[i].N, arr[i].K, arr[i].splitk) for i in range(n))
    if name in ("cg_norm_act_fwd_many", "cg_norm_act_bwd_many"):
        arr, n = args[0], (args[1] if name.endswith("fwd_many") else args[2])
        return " ".join("[%dx%dx%d%s%s%s]" % (arr[i].xv.n[0], arr[i].xv.n[1], arr[i].xv.n[2] * arr[i].xv.n[3], " bn" if arr[i].bn_mode else "",
                                              " pre" if arr[i].pre else "", " add" if arr[i].add else "") for i in range(n))
    if name.startswith("cg_dstd_tail"):
        t = ctypes.cast(args[0], ctypes.POINTER(_lib.DstdTail)).contents
        return "phase %d  B%d C%d T%d V%d" % (args[1], t.B, t.C, t.T, t.V)
    if name.startswith("cg_stgcn_domain"):
        return str(args[-7:-1] if name.endswith("fwd") else args[-8:-2])
    return ""


def call(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *args); e1.record()
    rows.append((name, describe(name, args), bench._call_bytes(name, args)[1], e0, e1))


_lib.call = call
step.replay()
torch.cuda.synchronize()
_lib.call = orig
agg = collections.OrderedDict()
for name, d, nb, e0, e1 in rows:
    a = agg.setdefault((name, d), [0, 0.0, 0])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += nb
tot = sum(a[1] for a in agg.values())
print("%s: %d calls, %.0f us summed" % (wl, len(rows), tot))
for (name, d), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.0f us %5.1f%% x%-3d %6.0f GB/s  %-24s %s" % (a[1], 100 * a[1] / tot, a[0], a[2] / max(a[1], 1e-9) / 1e3, name, d[:230]))
