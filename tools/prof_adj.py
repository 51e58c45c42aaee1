#!/usr/bin/env python3
"""GPU box: the phase kernels of the Map2Adj tail (csrc/map2adj_tail.hip) at one block shape, forward + backward, a few
repetitions: the program rocprofv3 is pointed at (tools/gpu_pmc_kernels.sh).  Usage: prof_adj.py [B T V reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
from cistgcn_amd import ops
from cistgcn_amd.models.CISTGCN.CISTGCN import Stage, _conv

B, T, V, reps = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (256, 50, 22, 3)
dev = "cuda"
exps = nn.ModuleList([Stage(s0=_conv(ch, ch), s1=nn.BatchNorm2d(ch), s3=nn.PReLU(), s4=_conv(ch, ch)) for ch in (V, T)]).to(dev).train()
R = lambda *s: torch.randn(*s, device=dev)
for _ in range(reps):
    s0, s1 = [R(B, V, T).requires_grad_(True) for _ in range(2)]
    q0, q1 = [R(B, T, V).requires_grad_(True) for _ in range(2)]
    ops.begin_step(torch.device(dev), bump_seed=True)
    adj = ops.map2adj_tail([(0, s0, q0), (1, s1, q1)], list(exps), True, drop_p=0.1, salts=(3, 4))
    torch.autograd.backward(list(adj), [torch.randn_like(a) for a in adj])
torch.cuda.synchronize()
print("done")
