#!/usr/bin/env python3
"""GPU box: R eager forward + backward passes of ONE DSTD_GC block (C -> C on (T, V), train mode, dropout as the benchmark) - the program
rocprofv3 is pointed at for the HBM traffic of a block invocation (FETCH_SIZE / WRITE_SIZE passes of tools/gpu_pmc_kernels.sh), the
denominator `roofline.traffic` of bench.py is compared with SURVEY 8(d)'s algorithmic bytes.  Usage: prof_block.py [C,B,T,V] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cistgcn_amd import ops
from cistgcn_amd.models import CISTGCN_0

C, B, T, V = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64,256,50,22").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CISTGCN_0(*bench.make_cfg(C, T, V, 0.1)).to(dev).train()
blk = net.st_gcnns[1]
x = torch.randn(B, C, T, V, device=dev, requires_grad=True)
for _ in range(reps):
    ops.begin_step(dev, bump_seed=False)
    net._site = 0
    x.grad = None
    y = net._block_staged(blk, x)
    y = y[0] if isinstance(y, tuple) else y
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
print("done", C, B, T, V, reps)
