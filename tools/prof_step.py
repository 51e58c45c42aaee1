#!/usr/bin/env python3
"""GPU box: a few eager training steps of a bench workload - the program rocprofv3 is pointed at for per-kernel PMC
counters of the whole step (tools/gpu_pmc_kernels.sh).  Usage: prof_step.py [workload] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cistgcn_amd.models import CISTGCN_0
from cistgcn_amd.runtime import EagerStep

wl = sys.argv[1] if len(sys.argv) > 1 else bench.HEADLINE
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
C, B, T, V = bench.WORKLOADS[wl]
torch.manual_seed(0)
net = CISTGCN_0(*bench.make_cfg(C, T, V, 0.1)).cuda().train()
x, tgt = bench.synth(B, T, V, 0)
step = EagerStep(net, x.cuda(), tgt.cuda())
for _ in range(steps):
    step.replay()
torch.cuda.synchronize()
print("done", wl, steps)
