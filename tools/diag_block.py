#!/usr/bin/env python3
"""Diagnostic (GPU box): one DSTD block in isolation, HIP vs fp64/fp32 CPU oracle, per-parameter errors."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import checks
from cistgcn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = sys.argv[2] if len(sys.argv) > 2 else "o"
g = torch.Generator().manual_seed(7)
net, ora = checks.build_pair(8, 10, 22, "cuda")
with torch.no_grad():
    for p in ora.parameters():
        p.add_(0.3 * torch.randn(p.shape, generator=g) / max(1.0, float(p[0].numel()) ** 0.5 if p.dim() > 1 else 3.0))
net.load_state_dict(ora.state_dict())
ora.train(); net.train()
ora64 = copy.deepcopy(ora).double()
if which == "o":
    shape, get = (B, 3, 22, 25), (lambda m: m.st_gcnns_o[0])
else:
    i = int(which)
    cin = 10 if i == 0 else 8
    shape, get = (B, cin, 10, 22), (lambda m: m.st_gcnns[i])
x = torch.randn(*shape, generator=g) * 2 + 0.5
strided = len(sys.argv) > 3 and sys.argv[3] == "strided"
gy = None
outs = {}
for name, model, xx in (("f64", ora64, x.double()), ("f32", ora, x.clone()), ("hip", net, x.clone().cuda())):
    xx.requires_grad_(True)
    if name == "hip":
        ops.begin_step("cuda")
        xin = xx
        if strided:     # same values through a permuted view, as the output block sees x7 (CISTGCN.py:592)
            base = xx.detach().permute(0, 3, 2, 1).contiguous().requires_grad_(True)
            xin = base.permute(0, 3, 2, 1)
        y = model._block(get(model), xin)
    else:
        y = model.block(get(model), xx)
    if gy is None:
        gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy.to(y.dtype).to(y.device))
    if name == "hip" and strided:
        xx.grad = base.grad.permute(0, 3, 2, 1)
    outs[name] = (y.detach().cpu().double(), xx.grad.cpu().double(), {k: p.grad.cpu().double() for k, p in get(model).named_parameters() if p.grad is not None})
r = outs["f64"]
print("y : hip %.2e cpu %.2e |ref| %.2e" % ((outs["hip"][0] - r[0]).abs().max(), (outs["f32"][0] - r[0]).abs().max(), r[0].abs().max()))
print("dx: hip %.2e cpu %.2e |ref| %.2e" % ((outs["hip"][1] - r[1]).abs().max(), (outs["f32"][1] - r[1]).abs().max(), r[1].abs().max()))
for k, ref in r[2].items():
    eh = float((outs["hip"][2][k] - ref).abs().max()); ec = float((outs["f32"][2][k] - ref).abs().max())
    print("%s %-45s hip %.2e cpu %.2e ratio %8.1f |ref| %.2e" % ("*" if eh > 8 * ec and eh > 1e-5 * float(ref.abs().max()) else " ", k, eh, ec, eh / max(ec, 1e-30), float(ref.abs().max())))
