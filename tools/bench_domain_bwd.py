#!/usr/bin/env python3
"""GPU box: time the fused ST-GCN stage kernels (forward and backward, both domains) at the shapes of one forward of a
workload, HIP events over repeated launches.  CG_DOM_BWD_VALU=1 selects the VALU backward for an A/B in one process run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cistgcn_amd import _lib, ops

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

shapes = [(256, 64, 64, 50, 22), (256, 10, 64, 50, 22), (256, 64, 10, 50, 22), (256, 32, 32, 50, 25), (16, 64, 64, 50, 22), (256, 64, 64, 10, 22)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, ci, co, T, V) in shapes:
    for dom in (0, 1):
        x = torch.randn(B, ci, T, V, device="cuda")
        adj = torch.randn((B, V, T, T) if dom == 0 else (B, T, V, V), device="cuda") * 0.1
        w = torch.randn(co, ci, device="cuda") * 0.1
        b = torch.randn(co, device="cuda")
        y = torch.empty(B, co, T, V, device="cuda")
        dy = torch.randn(B, co, T, V, device="cuda")
        dx, dadj, dw, db = torch.empty_like(x), torch.empty_like(adj), torch.empty_like(w), torch.empty_like(b)
        ws = torch.zeros(int(_lib.lib().cg_stgcn_domain_bwd_ws_floats(ci, co)), device="cuda")
        st = ops._stream(x)
        p = ops._ptr
        tf = timeit(lambda: _lib.call("cg_stgcn_domain_fwd", p(x), p(adj), p(w), p(b), p(y), None, B, ci, co, T, V, dom, st))
        res = {}
        for name, env in (("mfma", None), ("valu", "1")):
            if env: os.environ["CG_DOM_BWD_VALU"] = env
            else: os.environ.pop("CG_DOM_BWD_VALU", None)
            res[name] = timeit(lambda: _lib.call("cg_stgcn_domain_bwd", p(x), p(adj), p(w), p(dy), p(dx), p(dadj), p(dw), p(db), p(ws),
                                                 B, ci, co, T, V, dom, 0, st))
            res[name + "_dx"] = dx.clone(); res[name + "_dw"] = dw.clone(); res[name + "_da"] = dadj.clone()
        os.environ.pop("CG_DOM_BWD_VALU", None)
        ng, j = (V, T) if dom == 0 else (T, V)
        bytes_b = 4.0 * (2 * B * ci * T * V + 2 * B * ng * j * j + B * co * T * V)
        flops_b = 2.0 * B * (3 * ci * ng * j * j + 2 * ci * co * T * V)
        err = max(float((res["mfma_dx"] - res["valu_dx"]).abs().max() / res["valu_dx"].abs().max()),
                  float((res["mfma_dw"] - res["valu_dw"]).abs().max() / res["valu_dw"].abs().max()),
                  float((res["mfma_da"] - res["valu_da"]).abs().max() / res["valu_da"].abs().max()))
        print("B%d %d->%d T%d V%d dom%d | fwd %.1f us | bwd mfma %.1f us (%.0f GB/s, %.1f TF)  valu %.1f us | rel diff %.1e" % (
            B, ci, co, T, V, dom, tf, res["mfma"], bytes_b / res["mfma"] / 1e3, flops_b / res["mfma"] / 1e6, res["valu"], err), flush=True)
