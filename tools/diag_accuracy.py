#!/usr/bin/env python3
"""Diagnostic (GPU box): per-parameter gradient error of the HIP path vs an fp64 oracle run, next to the
fp32 CPU oracle's own error, in execution order.  Usage: diag_accuracy.py C T V B [fused]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import checks
from cistgcn_amd import ops
from oracle import cistgcn_ref as O

C, T, V, B = [int(a) for a in sys.argv[1:5]]
fused = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
g = torch.Generator().manual_seed(1000)
net, ora = checks.build_pair(C, T, V, "cuda", fused=fused)
with torch.no_grad():
    for p in ora.parameters():
        p.add_(0.3 * torch.randn(p.shape, generator=g) / max(1.0, float(p[0].numel()) ** 0.5 if p.dim() > 1 else 3.0))
net.load_state_dict(ora.state_dict())
x = 50 + 350 * torch.randn(B, T, V, 3, generator=g)
tgt = x[:, -1:] + 20 * torch.randn(B, 25, V, 3, generator=g)
ora.train(); net.train()
ora64 = copy.deepcopy(ora).double()
xo, x64, xd = x.clone().requires_grad_(True), x.double().requires_grad_(True), x.clone().cuda().requires_grad_(True)
po, = ora(xo); p64, = ora64(x64); pd, = net(xd)
O.mpjpe(po, tgt).backward(); O.mpjpe(p64, tgt.double()).backward(); ops.mpjpe(pd, tgt.cuda()).backward()
print("pred: hip %.3e cpu %.3e |ref| %.1f" % ((pd.detach().cpu().double() - p64).abs().max(), (po.double() - p64).abs().max(), p64.abs().max()))
print("dx  : hip %.3e cpu %.3e |ref| %.3e" % ((xd.grad.cpu().double() - x64.grad).abs().max(), (xo.grad.double() - x64.grad).abs().max(), x64.grad.abs().max()))
def attr(n, k):
    o = n
    for part in k.split("."):
        o = o[int(part)] if part.isdigit() else getattr(o, part)
    return o
for k in ["st_gcnns.0.w1", "st_gcnns.0.dsgn.Adj", "st_gcnns.0.tsgn.Adj", "st_gcnns.2.dsgn.Adj", "st_gcnns.4.tsgn.Adj", "context_layer.joints",
          "context_layer.seq_joints_dims", "st_gcnns_o.0.w1", "st_gcnns_o.0.dsgn.Adj"]:
    r = attr(ora64, k).detach()
    print("%-32s hip %.3e cpu %.3e |ref| %.3e" % (k, (attr(net, k).detach().cpu().double() - r).abs().max(), (attr(ora, k).detach().double() - r).abs().max(), r.abs().max()))
gd, gc = dict(net.named_parameters()), dict(ora.named_parameters())
rows = []
for k, p in ora64.named_parameters():
    r = p.grad
    eh = float((gd[k].grad.cpu().double() - r).abs().max()); ec = float((gc[k].grad.double() - r).abs().max())
    rows.append((k, eh, ec, float(r.abs().max())))
bad = [r for r in rows if r[1] > 8 * r[2] and r[1] > 1e-4 * max(1e-2, r[3])]
print("params failing the 8x / 1e-4 rule: %d of %d" % (len(bad), len(rows)))
for k, eh, ec, m in rows:
    flag = "*" if (eh > 8 * ec and eh > 1e-4 * max(1e-2, m)) else " "
    if flag == "*" or eh > 4 * ec:
        print("%s %-55s hip %.2e cpu %.2e ratio %6.1f |ref| %.2e" % (flag, k, eh, ec, eh / max(ec, 1e-30), m))
