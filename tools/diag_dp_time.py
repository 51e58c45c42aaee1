#!/usr/bin/env python3
"""Where does a data-parallel step spend its time?  Single process (world size 1, gloo): times graph replay, the
gradient gather and the all-reduce of the flat buffer separately.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench
from cistgcn_amd import ops
from cistgcn_amd.models import CISTGCN_0
from cistgcn_amd.runtime import FlatGrads, GraphedStep, allreduce_mean_

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group("gloo", rank=0, world_size=1)
dev = torch.device("cuda", 0)
C, B, T, V = bench.WORKLOADS["cistgcn8_b16_t50_v22"]
torch.manual_seed(0)
net = CISTGCN_0(*bench.make_cfg(C, T, V, 0.1)).to(dev).train()
x, tgt = [t.to(dev) for t in bench.synth(B, T, V, 0)]
flat = FlatGrads(net.parameters(), dev)
step = GraphedStep(net, x, tgt, warmup=3, flat=flat)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("graph replay only      %.3f ms" % timed(step.graph.replay))
print("replay + gather        %.3f ms" % timed(step.replay))
print("gather only            %.3f ms" % timed(flat.gather))
print("all-reduce (gloo, GPU tensor, %d floats)  %.3f ms" % (flat.flat.numel(), timed(lambda: allreduce_mean_(flat.flat))))
cpu = flat.flat.cpu()
print("all-reduce (gloo, CPU tensor)  %.3f ms" % timed(lambda: dist.all_reduce(cpu)))
dist.destroy_process_group()
