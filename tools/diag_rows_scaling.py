#!/usr/bin/env python3
"""Diagnostic (GPU box): time of the frame-collapsing convolution (one workgroup per sample) against the batch size - does a
sample's workgroup run alone at its latency, or do more workgroups per CU overlap?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cistgcn_amd import ops
dev = torch.device("cuda:0")
for (C, O) in ((32, 32), (64, 64)):
    for B in (64, 128, 256, 512, 1024, 2048):
        x = torch.randn(B, C, 50, 22, device=dev)
        w = 0.1 * torch.randn(O, C, 50, device=dev)
        for _ in range(3):
            ops.begin_step(dev); ops.collapse_rows(x, w, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.begin_step(dev)
        e0.record()
        for _ in range(20):
            ops.collapse_rows(x, w, False)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print("C=%d O=%d B=%4d: %7.1f us per forward, %6.1f MB, %5.0f GB/s" % (C, O, B, us, x.numel() * 4 / 1e6, x.numel() * 4 / us / 1e3))
