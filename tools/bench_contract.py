#!/usr/bin/env python3
"""Kernel-only timing of the contraction shapes that dominate a training step (GPU box).  Each case is planned once
and its descriptor relaunched back to back between HIP events, so host time and zero-fills are excluded.
Usage: bench_contract.py [C B T V]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cistgcn_amd import ops, _lib

C, B, T, V = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (64, 256, 50, 22)
dev = torch.device("cuda", 0)
mid = max(1, C // 2)
R = lambda *s: torch.randn(*s, device=dev)
x, xm, x2 = R(B, C, T, V), R(B, mid, T, V), R(B, 2 * C, T, V)
adjs, adjt = R(B, V, T, T), R(B, T, V, V)
cases = [
    ("pw fwd C->C", "oc,bchw->bohw", R(C, C), x),
    ("pw fwd 2C->C", "oc,bchw->bohw", R(C, 2 * C), x2),
    ("pw dX C->2C", "oc,bohw->bchw", R(C, 2 * C), x),
    ("pw dW CxC", "bohw,bchw->oc", x, x.clone()),
    ("pw dW Cx2C", "bohw,bchw->oc", x, x2),
    ("tower VxV fwd (B,V,T,T)", "oc,bchw->bohw", R(V, V), adjs),
    ("tower VxV dX", "oc,bohw->bchw", R(V, V), adjs),
    ("tower VxV dW", "bohw,bchw->oc", adjs, adjs.clone()),
    ("tower TxT fwd (B,T,V,V)", "oc,bchw->bohw", R(T, T), adjt),
    ("tower TxT dW", "bohw,bchw->oc", adjt, adjt.clone()),
    ("rows (T,1) C->mid", "och,bchw->bow", R(mid, C, T), x),
    ("rows dW", "bow,bchw->och", R(B, mid, V), x),
    ("rows dX", "och,bow->bchw", R(mid, C, T), R(B, mid, V)),
    ("cols (1,V) C->mid", "ocw,bchw->boh", R(mid, C, V), x),
    ("matvec space", "bvtx,bxv->bvt", R(B, V, T, T), R(B, T, V)),
    ("outer space", "bvt,bxv->bvtx", R(B, V, T), R(B, T, V)),
]
REP = 20
tot = 0.0
for name, spec, a, b in cases:
    probe = ops._contract_launch([lambda out, spec=spec, a=a, b=b: ops._contract_prepare(spec, a, b, out=out)], dev)
    d = probe[0].desc
    arr = (_lib.ContractDesc * 1)(d)
    st = ops._stream(a)
    for _ in range(3):
        _lib.call("cg_contract_many", arr, 1, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        _lib.call("cg_contract_many", arr, 1, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REP * 1e3
    nbytes = 4.0 * (a.numel() + b.numel() + probe[0].y.numel())
    tot += us
    print("%-28s %-16s G%-5d M%-5d N%-7d K%-7d sk%-4d %8.1f us %7.1f GB/s" % (name, spec, d.G, d.M, d.N, d.K, d.splitk, us, nbytes / us / 1e3))
print("sum %.1f us" % tot)
