#!/usr/bin/env python3
"""Row-kernel throughput on one large tensor (GPU box): BatchNorm(train)+PReLU forward and backward, with and without a
residual addend and dropout; achieved GB/s from the algorithmic traffic of each pass.  Usage: bench_rows.py [B C T V]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn as nn
from cistgcn_amd import ops

B, C, T, V = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (256, 64, 50, 22)
dev = torch.device("cuda", 0)
x = torch.randn(B, C, T, V, device=dev, requires_grad=True)
add = torch.randn(B, C, T, V, device=dev, requires_grad=True)
bn, pr = nn.BatchNorm2d(C).to(dev), nn.PReLU().to(dev)
mb = x.numel() * 4 / 1e6
def timed(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, kw, rd_f, wr_f, rd_b, wr_b in (("BN+PReLU", {}, 1, 1, 4, 1), ("BN+add+PReLU", {"add": add}, 2, 1, 6, 2),
                                          ("BN+drop+add+PReLU", {"add": add, "drop_p": 0.1, "salt": 3}, 2, 1, 6, 2)):
    def fwd():
        ops.begin_step(dev)
        st = ops._arena(dev).take(2 * C * 16)
        return ops.norm_act(x, bn=bn, train=True, prelu=pr, **kw)
    # forward incl. separate statistics pass (the model gets the sums from the producer's epilogue)
    tf = timed(fwd)
    y = fwd(); gy = torch.randn_like(y)
    def both():
        x.grad = None; add.grad = None
        y = fwd(); y.backward(gy)
    tb = timed(both) - tf
    print("%-20s fwd(+stats pass) %7.1f us   bwd (reduce+apply) %7.1f us = %6.0f GB/s  [tensor %.0f MB]" % (
        name, tf, tb, (rd_b + wr_b) * mb / tb * 1e3 if tb > 0 else 0, mb))
