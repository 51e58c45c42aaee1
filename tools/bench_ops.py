#!/usr/bin/env python3
"""Micro-benchmark (GPU box): representative contractions / row problems of one training step, timed alone with
HIP events.  Usage: bench_ops.py [C B T V]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cistgcn_amd import ops

C, B, T, V = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (64, 256, 50, 22)
dev = torch.device("cuda", 0)
mid = max(1, C // 2)
x = torch.randn(B, C, T, V, device=dev)
dy = torch.randn(B, mid, T, V, device=dev)
dyc = torch.randn(B, C, T, V, device=dev)
o = torch.randn(B, V, T, T, device=dev)
cases = [
    ("pw C->mid fwd", "oc,bchw->bohw", torch.randn(mid, C, device=dev), x),
    ("pw C->C fwd", "oc,bchw->bohw", torch.randn(C, C, device=dev), x),
    ("pw dX mid->C", "oc,bohw->bchw", torch.randn(mid, C, device=dev), dy),
    ("pw dW (mid x C)", "bohw,bchw->oc", dy, x),
    ("pw dW (C x C)", "bohw,bchw->oc", dyc, x),
    ("rows (T,1) C->mid", "och,bchw->bow", torch.randn(mid, C, T, device=dev), x),
    ("cols (1,V) C->mid", "ocw,bchw->boh", torch.randn(mid, C, V, device=dev), x),
    ("expansor V->V on (B,V,T,T)", "oc,bchw->bohw", torch.randn(V, V, device=dev), o),
    ("outer space", "bvt,bxv->bvtx", torch.randn(B, V, T, device=dev), torch.randn(B, T, V, device=dev)),
    ("compressor 2C->C", "oc,bchw->bohw", torch.randn(C, 2 * C, device=dev), torch.randn(B, 2 * C, T, V, device=dev)),
]
for name, spec, a, b in cases:
    for _ in range(3):
        y = ops._contract_raw(spec, a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        y = ops._contract_raw(spec, a, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    ins, ly = spec.split("->"); la, lx = ins.split(",")
    sizes = {}
    for t, ls in ((a, la), (b, lx)):
        sizes.update({l: n for l, n in zip(ls, t.shape)})
    flops = 2.0
    for l in set(la + lx):
        flops *= sizes[l]
    nbytes = 4.0 * (a.numel() + b.numel() + y.numel())
    print("%-30s %8.1f us  %7.2f TFLOP/s  %7.1f GB/s   out %s" % (name, us, flops / us / 1e6, nbytes / us / 1e3, tuple(y.shape)))
# row kernels
import torch.nn as nn
bn = nn.BatchNorm2d(C).to(dev); pr = nn.PReLU().to(dev)
xx = x.clone().requires_grad_(True)
for name, fn in [("norm_act fwd (BN train + PReLU)", lambda: ops.norm_act(x, bn=bn, train=True, prelu=pr))]:
    for _ in range(3):
        ops.begin_step(dev); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.begin_step(dev); fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%-30s %8.1f us  (stats + apply + arena memset; tensor %.0f MB)" % (name, us, x.numel() * 4 / 1e6))
