#!/usr/bin/env python3
"""Diagnostic (GPU box): time of the BatchNorm row kernel at a few channel counts of a (256, C, 50, 22) tensor, train mode,
back-to-back launches between two events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
from cistgcn_amd import ops
dev = torch.device("cuda:0")
for C in (10, 16, 10, 32, 64):
    x = (50 + 350 * torch.randn(256, C, 50, 22, device=dev))
    bn = nn.BatchNorm2d(C).to(dev).train()
    for _ in range(3):
        ops.begin_step(dev)
        ops.norm_act(x, bn=bn, train=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.begin_step(dev)
    e0.record()
    for _ in range(10):
        ops.norm_act(x, bn=bn, train=True)
    e1.record(); torch.cuda.synchronize()
    print("C=%d: %.1f us per norm_act (stats + apply), %.1f MB" % (C, e0.elapsed_time(e1) * 100, x.numel() * 4 / 1e6))
