#!/usr/bin/env python3
"""GPU box: the phase kernels of the DSTD_GC tail (csrc/dstd_tail.hip) at one block shape, forward + backward, a few
repetitions: the program rocprofv3 is pointed at (tools/gpu_pmc_kernels.sh).  Usage: prof_tail.py [B C T V reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
from cistgcn_amd import ops
from cistgcn_amd.models.layers.SE import SELayer2d

B, C, T, V, reps = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (256, 64, 50, 22, 3)
dev = "cuda"
bns = nn.ModuleList([nn.BatchNorm2d(C) for _ in range(5)]).to(dev).train()
al = nn.ModuleList([nn.PReLU() for _ in range(5)]).to(dev)
conv = nn.Conv2d(2 * C, C, 1, bias=False).to(dev)
se = SELayer2d(C, reduction=8).to(dev)
R = lambda *s: torch.randn(*s, device=dev)
for _ in range(reps):
    y1, y2, r1, r2, bres = [R(B, C, T, V).requires_grad_(True) for _ in range(5)]
    w1, w2 = [R(B, C).requires_grad_(True) for _ in range(2)]
    ops.begin_step(torch.device(dev), bump_seed=True)
    def sums(y):
        st = ops._arena(torch.device(dev, 0)).take(2 * C * 16)
        yc = y.detach().double()
        st.view(16, C, 2)[0].copy_(torch.stack((yc.sum((0, 2, 3)), (yc * yc).sum((0, 2, 3))), 1))
        return st
    out, _ = ops.dstd_tail([y1, y2], [sums(y1), sums(y2)], [r1, r2], (w1, w2), list(bns), list(al), conv.weight, se, bres, True,
                           drop_p=0.1, salts=(3, 4), emit_stats=True)
    out.backward(torch.randn_like(out))
torch.cuda.synchronize()
print("done")
