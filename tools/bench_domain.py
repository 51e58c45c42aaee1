#!/usr/bin/env python3
"""Micro-benchmark (GPU box) of the fused ST-GCN stage: forward and backward, both domains, the hidden-block shape
of a workload, under the tile-geometry overrides CG_DOM_GT / CG_DOM_PER.  Usage: bench_domain.py C B T V"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cistgcn_amd import ops

C, B, T, V = [int(a) for a in sys.argv[1:5]]
dev = torch.device("cuda", 0)
def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for dom in (0, 1):
    x = torch.randn(B, C, T, V, device=dev, requires_grad=True)
    adj = (torch.randn(B, V, T, T, device=dev) if dom == 0 else torch.randn(B, T, V, V, device=dev)).mul_(0.1).requires_grad_(True)
    w = (torch.randn(C, C, device=dev) * 0.1).requires_grad_(True)
    b = torch.randn(C, device=dev, requires_grad=True)
    nb = 4.0 * (2 * x.numel() + adj.numel())
    f = timeit(lambda: ops.stgcn_domain(x.detach(), adj.detach(), w.detach(), b.detach(), dom))
    y, _ = ops.stgcn_domain(x, adj, w, b, dom)
    gy = torch.randn_like(y)
    def bwd():
        torch.autograd.grad(y, (x, adj, w, b), gy, retain_graph=True)
    t = timeit(bwd)
    print("GT=%s PER=%s dom %d: fwd %7.1f us (%6.0f GB/s)   bwd %7.1f us" % (os.environ.get("CG_DOM_GT", "-"), os.environ.get("CG_DOM_PER", "-"), dom, f, nb / f / 1e3, t))
