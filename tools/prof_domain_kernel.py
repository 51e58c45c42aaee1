#!/usr/bin/env python3
"""Launches only the dominant kernel (fused ST-GCN stage, forward, space domain) on the six shapes it takes in one
forward of the given workload, R times each - the population bench.py's roofline block times.  Run under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) to obtain HBM traffic per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cistgcn_amd import ops

w = sys.argv[1] if len(sys.argv) > 1 else "cistgcn8_b16_t50_v22"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
C, B, T, V = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
for (ci, co, t, v) in bench.domain_shapes(C, T, V):
    x = torch.randn(B, ci, t, v, device=dev)
    adj = torch.randn(B, v, t, t, device=dev) * 0.1
    wt = torch.randn(co, ci, device=dev) * 0.1
    b = torch.randn(co, device=dev)
    for _ in range(reps):
        ops.stgcn_domain(x, adj, wt, b, 0)
torch.cuda.synchronize()
print("done", w, reps)
