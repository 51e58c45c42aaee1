#!/usr/bin/env python3
"""Diagnostic: where a workgroup of the stacked pointwise maps' backward spends its cycles (private stamped library,
`python tools/stamps_planes.py --build`).  Usage: stamps_pwm.py [B Cin T V M1,M2,..]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "libcistgcn_stamps.so")
import torch
from cistgcn_amd import _lib
_lib.LIB_PATH = OUT
from cistgcn_amd import ops
h = _lib.lib()
B, C, T, V = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (256, 64, 50, 22)
Ms = [int(m) for m in sys.argv[5].split(",")] if len(sys.argv) > 5 else [32, 32, 32, 32]
dev = "cuda"
buf = torch.zeros(1024 * 256, dtype=torch.int64, device=dev)
h.cg_pwm_set_stamps.argtypes = [ctypes.c_void_p]


def run():
    x = torch.randn(B, C, T, V, device=dev, requires_grad=True)
    ws = [(0.1 * torch.randn(M, C, device=dev)).requires_grad_(True) for M in Ms]
    ops.begin_step(torch.device(dev))
    ys = [o[0] for o in ops.pointwise_maps(x, ws, True)]
    torch.autograd.backward(ys, [torch.randn_like(y) for y in ys])
    torch.cuda.synchronize()


run(); run()
assert h.cg_pwm_set_stamps(buf.data_ptr()) == 0
run()
st = buf.view(1024, 256).cpu()
n = int(st[:, 0].max())
rows = st[st[:, 0] == n][:, 1:n + 1].double()
d = rows[:, 1:] - rows[:, :-1]
m = d.mean(0)
tiles = (n - 4) // 4
print("workgroups %d, stamps %d, cycles per workgroup %.0f, tiles per workgroup %d" % (rows.shape[0], n, float((rows[:, -1] - rows[:, 0]).mean()), tiles))
print("prologue %.0f" % float(m[0]))
# per tile: A after the top barrier | B after commit + barrier | C after the prefetch issue | D after dW; dx + the next top barrier end at the next A
acc = [0.0] * 4
for t in range(tiles):
    for j in range(4):
        acc[j] += float(m[1 + 4 * t + j])
acc[0] += float(m[1 + 4 * tiles])          # dx of the last tile ends at the stamp in front of the epilogue
print("mean cycles per tile: top barrier (+ dx of the previous tile) %.0f | commit + barrier %.0f | prefetch issue %.0f | dW %.0f" % tuple(a / tiles for a in acc))
print("epilogue %.0f" % float(m[2 + 4 * tiles:].sum()))
