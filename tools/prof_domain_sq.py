#!/usr/bin/env python3
"""Fused ST-GCN stage on the 64->64 shape of the C=64 / B=256 workload, forward and backward, both domains, R times:
the launch population for SQ counter passes (`rocprofv3 --pmc ...`, tools/gpu_sq.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cistgcn_amd import ops

B, C, T, V = 256, 64, 50, 22
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
for dom in (0, 1):
    x = torch.randn(B, C, T, V, device=dev, requires_grad=True)
    adj = (torch.randn(B, V, T, T, device=dev) * 0.1 if dom == 0 else torch.randn(B, T, V, V, device=dev) * 0.1).requires_grad_(True)
    wt = (torch.randn(C, C, device=dev) * 0.1).requires_grad_(True)
    b = torch.randn(C, device=dev, requires_grad=True)
    for _ in range(reps):
        y, _ = ops.stgcn_domain(x, adj, wt, b, dom)
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
print("done")
