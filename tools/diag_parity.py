#!/usr/bin/env python3
"""Diagnostic (GPU box): per-tensor error table of the HIP model against the golden vectors / oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import checks
from helpers import load_case, state_of
from cistgcn_amd import ops
from oracle import cistgcn_ref as O

dev = sys.argv[1] if len(sys.argv) > 1 else "cuda"
name = sys.argv[2] if len(sys.argv) > 2 else "h36m_c8_t10_v22"
rec = load_case(name)
C, T, V, B = [int(v) for v in rec["meta"]]
for mode in ("eval", "train"):
    for fused in (True, False):
        net, ora = checks.build_pair(C, T, V, dev, state_of(rec), fused=fused)
        net.train(mode == "train"); ora.train(mode == "train")
        x = torch.from_numpy(rec["x"]).to(dev).requires_grad_(True)
        tgt = torch.from_numpy(rec["target"]).to(dev)
        pred, = net(x); loss = ops.mpjpe(pred, tgt); loss.backward()
        xo = torch.from_numpy(rec["x"]).requires_grad_(True)
        po, = ora(xo); lo = O.mpjpe(po, torch.from_numpy(rec["target"])); lo.backward()
        print("== %s fused=%s: pred err %.3e (|ref| %.1f)  loss %.6f vs %.6f  dx err %.3e (|ref| %.3e)" % (
            mode, fused, (pred.detach().cpu() - po).abs().max(), po.abs().max(), loss.item(), lo.item(),
            (x.grad.cpu() - xo.grad).abs().max(), xo.grad.abs().max()))
        rows = []
        go = dict(ora.named_parameters())
        for k, p in net.named_parameters():
            r = go[k].grad
            e = (p.grad.cpu() - r).abs().max().item()
            rows.append((e / max(1e-2, r.norm().item()), e, r.norm().item(), k))
        rows.sort(reverse=True)
        for rel, e, n, k in rows[:8]:
            print("   %-50s err %.3e  |ref| %.3e  rel %.3e" % (k, e, n, rel))
