#!/usr/bin/env python3
"""Diagnostic (GPU box): per PReLU site, forward error / sign mismatches / incoming-gradient error of the HIP
model vs the fp64 oracle (and the fp32 oracle for scale)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import checks
from cistgcn_amd import ops
from oracle import cistgcn_ref as O

C, T, V, B = 8, 10, 22, 8
g = torch.Generator().manual_seed(1000)
net, ora = checks.build_pair(C, T, V, "cuda")
with torch.no_grad():
    for p in ora.parameters():
        p.add_(0.3 * torch.randn(p.shape, generator=g) / max(1.0, float(p[0].numel()) ** 0.5 if p.dim() > 1 else 3.0))
net.load_state_dict(ora.state_dict())
x = 50 + 350 * torch.randn(B, T, V, 3, generator=g)
tgt = x[:, -1:] + 20 * torch.randn(B, 25, V, 3, generator=g)
ora.train(); net.train()
ora64 = copy.deepcopy(ora).double()

rec = {"f64": [], "f32": [], "hip": []}
cur = [None]
orig_act = O._act
def act_hook(xx, m):
    y = orig_act(xx, m)
    y.retain_grad()
    rec[cur[0]].append(y)
    return y
O._act = act_hook
orig_na = ops.norm_act
def na_hook(xx, **kw):
    y = orig_na(xx, **kw)
    if kw.get("prelu") is not None:
        y.retain_grad()
        rec["hip"].append((y, kw.get("add") if kw.get("add_post") else None))
    return y
ops.norm_act = na_hook

cur[0] = "f64"; x64 = x.double().requires_grad_(True); p64, = ora64(x64); O.mpjpe(p64, tgt.double()).backward()
cur[0] = "f32"; xo = x.clone().requires_grad_(True); po, = ora(xo); O.mpjpe(po, tgt).backward()
cur[0] = "hip"; xd = x.clone().cuda().requires_grad_(True); pd, = net(xd); ops.mpjpe(pd, tgt.cuda()).backward()
print(len(rec["f64"]), len(rec["f32"]), len(rec["hip"]))
for i, (r, c, (h, addp)) in enumerate(zip(rec["f64"], rec["f32"], rec["hip"])):
    hv = h.detach().cpu().double()
    if addp is not None:            # HIP site fuses the post-activation residual add; the oracle's _act does not
        hv = hv - addp.detach().cpu().double()
    if hv.shape != r.shape:
        print(i, "shape mismatch", tuple(hv.shape), tuple(r.shape)); continue
    r_ = r.detach()
    flips_h = int(((hv > 0) != (r_ > 0)).sum()); flips_c = int(((c.detach().double() > 0) != (r_ > 0)).sum())
    gh = h.grad.cpu().double(); gc = c.grad.double(); gr = r.grad
    print("%3d %-18s fwd hip %.1e cpu %.1e | flips hip %3d cpu %3d | dy hip %.2e cpu %.2e |dy| %.2e" % (
        i, str(tuple(r.shape)), (hv - r_).abs().max(), (c.detach().double() - r_).abs().max(), flips_h, flips_c,
        (gh - gr).abs().max(), (gc - gr).abs().max(), gr.abs().max()))
