#!/usr/bin/env python3
"""Diagnostic: where a workgroup of the plane backward spends its cycles.  Builds a PRIVATE copy of the library with
-DCG_DOMP_STAMPS (build/libcistgcn_stamps.so; `--build` on the CPU box, it travels with the snapshot), runs the kernel on the
GPU box and prints the mean cycles between consecutive stamps.  The shipped library contains no stamp."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "libcistgcn_stamps.so")
if "--build" in sys.argv:
    from cistgcn_amd import build
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCG_DOMP_STAMPS", "-DCG_TAIL_STAMPS", "-DCG_ADJ_STAMPS", "-I", build.CSRC, "-o", OUT] + build.sources()
    subprocess.check_call(cmd)
    print(OUT)
    sys.exit(0)
import torch
from cistgcn_amd import _lib
_lib.LIB_PATH = OUT
from cistgcn_amd import ops
h = _lib.lib()
B, ci, co, T, V = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "256,64,64,50,22").split(",")]
dom = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x = torch.randn(B, ci, T, V, device="cuda"); adj = torch.randn((B, V, T, T) if dom == 0 else (B, T, V, V), device="cuda") * 0.1
w = torch.randn(co, ci, device="cuda") * 0.1; dy = torch.randn(B, co, T, V, device="cuda")
dx, dadj, dw, db = torch.empty_like(x), torch.empty_like(adj), torch.empty_like(w), torch.empty(co, device="cuda")
ws = torch.zeros(int(h.cg_stgcn_domain_bwd_ws_floats(ci, co)), device="cuda")
nblk = 8 * ((B + 7) // 8) * 8 * 2
buf = torch.zeros(nblk * 256, dtype=torch.int64, device="cuda")
p = ops._ptr
run = lambda: _lib.call("cg_stgcn_domain_bwd", p(x), p(adj), p(w), p(dy), p(dx), p(dadj), p(dw), p(db), p(ws), B, ci, co, T, V, dom, 0, ops._stream(x))
for _ in range(3): run()
torch.cuda.synchronize()
h.cg_domp_set_stamps.argtypes = [ctypes.c_void_p]
assert h.cg_domp_set_stamps(buf.data_ptr()) == 0
run(); torch.cuda.synchronize()
st = buf.view(nblk, 256).cpu()
n = int(st[:, 0].max())
rows = st[st[:, 0] == n][:, 1:n + 1].double()
d = (rows[:, 1:] - rows[:, :-1])
print("workgroups %d, stamps %d, total cycles per workgroup: mean %.0f" % (rows.shape[0], n, float((rows[:, -1] - rows[:, 0]).mean())))
m = d.mean(0)
for i in range(n - 1):
    print("%3d -> %3d  %8.0f" % (i + 1, i + 2, float(m[i])))
