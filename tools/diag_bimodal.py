#!/usr/bin/env python3
"""The B=16 step is 5.30 or 5.55 ms depending on the process.  Is it the memory the graph was captured into (then a
re-capture in the same process changes it) or the process / device state (then it does not)?  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cistgcn_amd import ops
from cistgcn_amd.models import CISTGCN_0
from cistgcn_amd.runtime import GraphedStep

dev = torch.device("cuda", 0)
C, B, T, V = bench.WORKLOADS["cistgcn8_b16_t50_v22"]
x, tgt = [t.to(dev) for t in bench.synth(B, T, V, 0)]
def timed(step, n=50):
    for _ in range(10): step.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
keep = []
for cap in range(4):
    torch.manual_seed(0)
    net = CISTGCN_0(*bench.make_cfg(C, T, V, 0.1)).to(dev).train()
    ops.manual_seed(1234, dev)
    step = GraphedStep(net, x, tgt, warmup=3)
    print("capture %d: %s ms" % (cap, ["%.3f" % timed(step) for _ in range(3)]), flush=True)
    keep.append((net, step))                      # keep the memory of earlier captures alive: the next one lands elsewhere
    pad = torch.empty(int(3e6) * (cap + 1), device=dev)   # and shift the allocator
    keep.append(pad)
