#!/usr/bin/env python3
"""Diagnostic: where a workgroup of K3 (DSTD_GC tail, backward matrix phase) spends its cycles.  Uses the private stamped copy of
the library (build/libcistgcn_stamps.so, `python tools/stamps_planes.py --build` on the CPU box); prints the mean shader-clock
ticks (100 MHz) between consecutive stamps of the first tiles.  The shipped library contains no stamp.
Usage: stamps_tail.py [B C T V]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "libcistgcn_stamps.so")
import torch
import torch.nn as nn
from cistgcn_amd import _lib
_lib.LIB_PATH = OUT
from cistgcn_amd import ops
from cistgcn_amd.models.layers.SE import SELayer2d
h = _lib.lib()
B, C, T, V = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (256, 64, 50, 22)
dev = "cuda"
bns = nn.ModuleList([nn.BatchNorm2d(C) for _ in range(5)]).to(dev).train()
al = nn.ModuleList([nn.PReLU() for _ in range(5)]).to(dev)
conv = nn.Conv2d(2 * C, C, 1, bias=False).to(dev)
se = SELayer2d(C, reduction=8).to(dev)
R = lambda *s: torch.randn(*s, device=dev)
nblk = 4096
buf = torch.zeros(nblk * 256, dtype=torch.int64, device=dev)
h.cg_tail_set_stamps.argtypes = [ctypes.c_void_p]


def run():
    y1, y2, r1, r2, bres = [R(B, C, T, V).requires_grad_(True) for _ in range(5)]
    w1, w2 = [R(B, C).requires_grad_(True) for _ in range(2)]
    ops.begin_step(torch.device(dev), bump_seed=True)

    def sums(y):
        st = ops._arena(torch.device(dev, 0)).take(2 * C * 16)
        yc = y.detach().double()
        st.view(16, C, 2)[0].copy_(torch.stack((yc.sum((0, 2, 3)), (yc * yc).sum((0, 2, 3))), 1))
        return st
    out, _ = ops.dstd_tail([y1, y2], [sums(y1), sums(y2)], [r1, r2], (w1, w2), list(bns), list(al), conv.weight, se, bres, True,
                           drop_p=0.1, salts=(3, 4), emit_stats=True)
    out.backward(torch.randn_like(out))
    torch.cuda.synchronize()


run(); run()
assert h.cg_tail_set_stamps(buf.data_ptr()) == 0
run()
st = buf.view(nblk, 256).cpu()
n = int(st[:, 0].max())
rows = st[st[:, 0] == n][:, 1:n + 1].double()
d = rows[:, 1:] - rows[:, :-1]
print("workgroups %d, stamps %d, ticks per workgroup: mean %.0f (x24 = shader cycles at 2.4 GHz)" % (rows.shape[0], n, float((rows[:, -1] - rows[:, 0]).mean())))
m = d.mean(0)
# stamps: 1 start, 2 loop entry, per tile A (after the top barrier), B (after stage_act), C (after dh0 + barrier), D (after dWc);
# the d a product + g_p stores of a tile end at the next tile's A (or at the stamp in front of the final reduction)
print("prologue %.0f" % float(m[0]))
tiles = (n - 4) // 4
acc = [0.0] * 4
for t in range(tiles):
    for j in range(4):
        acc[j] += float(m[1 + 4 * t + j])
acc[0] += float(m[1 + 4 * tiles])
print("tiles per workgroup %d; mean ticks per tile: stage_act %.0f | dh0 + barrier %.0f | dWc %.0f | da + gp store + barrier %.0f" % (
    tiles, acc[1] / tiles, acc[2] / tiles, acc[3] / tiles, acc[0] / tiles))
print("tail (reduction, atomics) %.0f" % float(m[2 + 4 * tiles:].sum()))
