#!/usr/bin/env python3
"""GPU box: HIP-event time of the contraction launches behind ONE first-level map of a DSTD_GC block at a time (forward, weight
gradient, input gradient), B=256, C=64, T=50, V=22.  Usage: probe_maps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cistgcn_amd import _lib, ops

B, C, T, V = 256, 64, 50, 22
dev = "cuda"
x0 = torch.randn(B, C, T, V, device=dev)
cases = [("pointwise 64->32", "oc,bchw->bohw", (32, C)), ("pointwise 64->64", "oc,bchw->bohw", (64, C)),
         ("rows collapse (T,1) 64->32", "och,bchw->bow", (32, C, T)), ("cols collapse (1,V) 64->32", "ocw,bchw->boh", (32, C, V))]
orig = _lib.call
rows = []
def call(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(name, *args); e1.record()
    d = ""
    if name == "cg_contract_many":
        arr, n = args[0], args[1]
        d = " ".join("[m%d M%d N%d K%d s%d]" % (arr[i].mode, arr[i].M, arr[i].N, arr[i].K, arr[i].splitk) for i in range(n))
    rows.append((name, d, e0, e1))
for label, spec, wshape in cases:
    w = (0.1 * torch.randn(*wshape, device=dev)).requires_grad_(True)
    x = x0.clone().requires_grad_(True)
    for rep in range(3):
        ops.begin_step(torch.device(dev))
        rows.clear()
        _lib.call = call
        y = ops.contract(spec, w, x)
        gy = torch.randn_like(y)
        y.backward(gy)
        torch.cuda.synchronize()
        _lib.call = orig
        x.grad = None; w.grad = None
    print(label)
    for name, d, e0, e1 in rows:
        print("   %8.1f us  %-20s %s" % (e0.elapsed_time(e1) * 1e3, name, d))
