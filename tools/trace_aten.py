#!/usr/bin/env python3
"""Diagnostic (GPU box): which device kernels of one training step are NOT from libcistgcn_hip.so, and which Python
lines launch them (torch.profiler with stacks on an eager step)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from bench import make_cfg, synth
from cistgcn_amd import ops
from cistgcn_amd.models import CISTGCN_0
from cistgcn_amd.runtime import EagerStep

C, B, T, V = 8, 16, 50, 22
torch.manual_seed(0)
net = CISTGCN_0(*make_cfg(C, T, V, 0.1)).cuda().train()
x, tgt = synth(B, T, V, 0)
step = EagerStep(net, x.cuda(), tgt.cuda())
for _ in range(3):
    step.replay()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.replay()
    torch.cuda.synchronize()
by = collections.Counter()
evs = list(prof.events())
dev = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA]
names = collections.Counter(e.name[:70] for e in dev if "cg_" not in e.name[:48])
print("device events outside the library:")
for n, c in names.most_common():
    print("  %3d  %s" % (c, n))


def chain(ev):
    out, p = [], ev
    while p is not None and len(out) < 6:
        out.append(p.name[:48])
        p = p.cpu_parent
    return " < ".join(out)


for ev in evs:
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        continue
    ks = [k.name for k in getattr(ev, "kernels", [])]
    ks = [k for k in ks if "cg_" not in k[:48]]
    if not ks or any(c.kernels for c in ev.cpu_children if getattr(c, "kernels", None)):
        continue
    st = [s for s in (ev.stack or []) if "cistgcn_amd" in s or "bench.py" in s]
    by[(chain(ev), ks[0][:40], " <- ".join(s.split("/")[-1] for s in st[:3]))] += 1
for (op, k, where), n in by.most_common():
    print("%3d  %-110s %-40s %s" % (n, op, k, where))
print("total non-cg kernels:", sum(by.values()))
