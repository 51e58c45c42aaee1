#!/usr/bin/env python3
"""Diagnostic: where a workgroup of N2 (Map2Adj tail, second backward phase) spends its cycles, per tower.  Uses the private stamped
copy of the library (build/libcistgcn_stamps.so, `python tools/stamps_planes.py --build` on the CPU box); prints the mean
shader-clock ticks (100 MHz) between consecutive stamps.  The shipped library contains no stamp.
Usage: stamps_adj.py [B T V]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "libcistgcn_stamps.so")
import torch
import torch.nn as nn
from cistgcn_amd import _lib
_lib.LIB_PATH = OUT
from cistgcn_amd import ops
from cistgcn_amd.models.CISTGCN.CISTGCN import Stage, _conv
h = _lib.lib()
B, T, V = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (256, 50, 22)
dev = "cuda"
exps = nn.ModuleList([Stage(s0=_conv(ch, ch), s1=nn.BatchNorm2d(ch), s3=nn.PReLU(), s4=_conv(ch, ch)) for ch in (V, T)]).to(dev).train()
R = lambda *s: torch.randn(*s, device=dev)
nblk = 8192
buf = torch.zeros(nblk * 256, dtype=torch.int64, device=dev)
h.cg_adj_set_stamps.argtypes = [ctypes.c_void_p]


def run():
    s0, s1 = [R(B, V, T).requires_grad_(True) for _ in range(2)]
    q0, q1 = [R(B, T, V).requires_grad_(True) for _ in range(2)]
    ops.begin_step(torch.device(dev), bump_seed=True)
    adj = ops.map2adj_tail([(0, s0, q0), (1, s1, q1)], list(exps), True, drop_p=0.1, salts=(3, 4))
    torch.autograd.backward(list(adj), [torch.randn_like(a) for a in adj])
    torch.cuda.synchronize()


run(); run()
assert h.cg_adj_set_stamps(buf.data_ptr()) == 0
run()
st = buf.view(nblk, 256).cpu()
t0 = st[:, 1][st[:, 0] > 0]
print("kernel span: %.0f ticks (x24 = shader cycles at 2.4 GHz) over %d workgroups" % (float(st[:, 1:][st[:, 0] > 0].max() - t0.min()), int((st[:, 0] > 0).sum())))
for n in sorted(set(int(v) for v in st[:, 0].unique()) - {0}):
    rows = st[st[:, 0] == n][:, 1:n + 1].double()
    d = rows[:, 1:] - rows[:, :-1]
    m = d.mean(0)
    tiles = (n - 4) // 8
    print("== workgroups with %d stamps (%d tiles): %d workgroups, lifetime mean %.0f ticks" % (n, tiles, rows.shape[0], float((rows[:, -1] - rows[:, 0]).mean())))
    print("   prologue (tables, weights, constants, first fetch issue): %.0f" % float(m[0]))
    acc = [0.0] * 8
    for t in range(tiles):
        for j in range(8):
            acc[j] += float(m[1 + 8 * t + j])
    lab = ["E'->A dQ / dS cells of the previous tile + top barrier", "A->A1 wait for the prefetched tile, commit de", "A1->A2 seed image", "A2->A3 barrier", "A3->B prefetch issue", "B->C dW0 product", "C->D barrier", "D->E do product + image + barrier"]
    for j in range(8):
        print("   %-50s %8.1f per tile" % (lab[j], acc[j] / max(tiles, 1)))
    print("   last cells + barrier %.0f | epilogue: part + dW atomics %.0f" % (float(m[1 + 8 * tiles]), float(m[-1])))
