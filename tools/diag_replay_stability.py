#!/usr/bin/env python3
"""Replay stability of the captured training step: eager steps first (allocator and autograd history), then capture, then
many replays - with dropout 0 the loss must stay at the eager value (up to the order of fp32 atomics).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import checks
from cistgcn_amd import ops
from cistgcn_amd.runtime import GraphedStep, _drop_graph_attributes

worst = 0.0
for (C, T, V, B) in ((8, 10, 22, 4), (8, 50, 22, 16), (16, 10, 18, 6)):
    net, _ = checks.build_pair(C, T, V, "cuda"); net.train(); net.dropout = 0.0
    g = torch.Generator().manual_seed(5)
    x = (50 + 350 * torch.randn(B, T, V, 3, generator=g)).cuda()
    tgt = (x[:, -1:].cpu() + 20 * torch.randn(B, 25, V, 3, generator=g)).cuda()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    ref = None
    for i in range(3):
        net.zero_grad(set_to_none=True)
        pred, = net(x); l = ops.mpjpe(pred, tgt); l.backward(); ref = float(l.detach())
    gref = {k: p.grad.clone() for k, p in net.named_parameters()}
    del l, pred
    net.zero_grad(set_to_none=True); _drop_graph_attributes(net); net.load_state_dict(sd)
    step = GraphedStep(net, x, tgt, warmup=2)
    vals = []
    for i in range(120):
        step.replay()
        if i % 20 == 0:
            torch.cuda.synchronize(); vals.append(float(step.loss))
    if os.environ.get('DIAG_SYNC', '1') == '1':
        torch.cuda.synchronize()
    errs = sorted(((float((p.grad - gref[k]).abs().max()) / max(1e-6, float(gref[k].abs().max())), k, tuple(p.shape), p.grad.data_ptr()) for k, p in net.named_parameters()), reverse=True)
    gerr = errs[0][0]
    pool = ops._zero_pools[ops._dev("cuda")]
    lo, hi = pool.buf.data_ptr(), pool.buf.data_ptr() + pool.buf.numel() * 4
    if errs[0][0] > 1e3:
        print("   GARBAGE:", [(k, sh, "%.1e" % e, "in-pool" if lo <= ptr < hi else "own") for e, k, sh, ptr in errs if e > 1e3][:12])
    print("   worst:", [(k, sh, "%.1e" % e, "in-pool" if lo <= ptr < hi else "own") for e, k, sh, ptr in errs[:5]], " #bad(>1e-3):", sum(1 for e in errs if e[0] > 1e-3), "of", len(errs))
    dev = max(abs(v - ref) / abs(ref) for v in vals)
    worst = max(worst, dev)
    print((C, T, V, B), "eager %.6f" % ref, "replays", ["%.6f" % v for v in vals], "max rel dev %.2e" % dev, "grad rel dev %.2e" % gerr)
print("WORST", worst)
