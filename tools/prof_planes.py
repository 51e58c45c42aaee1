#!/usr/bin/env python3
"""Launches the fused ST-GCN stage (forward + backward, both domains) at one shape, R times: the population the rocprofv3 passes
of tools/gpu_pmc_kernels.sh read.  Usage: prof_planes.py [B,Cin,Cout,T,V] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cistgcn_amd import _lib, ops

B, ci, co, T, V = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,64,64,50,22").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = ops._ptr
for dom in (0, 1):
    x = torch.randn(B, ci, T, V, device="cuda")
    adj = torch.randn((B, V, T, T) if dom == 0 else (B, T, V, V), device="cuda") * 0.1
    w = torch.randn(co, ci, device="cuda") * 0.1
    b = torch.randn(co, device="cuda")
    y = torch.empty(B, co, T, V, device="cuda")
    dy = torch.randn(B, co, T, V, device="cuda")
    dx, dadj, dw, db = torch.empty_like(x), torch.empty_like(adj), torch.empty_like(w), torch.empty_like(b)
    ws = torch.zeros(int(_lib.lib().cg_stgcn_domain_bwd_ws_floats(ci, co)), device="cuda")
    st = ops._stream(x)
    for _ in range(reps):
        _lib.call("cg_stgcn_domain_fwd", p(x), p(adj), p(w), p(b), p(y), None, B, ci, co, T, V, dom, st)
        _lib.call("cg_stgcn_domain_bwd", p(x), p(adj), p(w), p(dy), p(dx), p(dadj), p(dw), p(db), p(ws), B, ci, co, T, V, dom, 0, st)
torch.cuda.synchronize()
print("done")
