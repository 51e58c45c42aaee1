#!/usr/bin/env python3
"""GPU box: A/B of the fused ST-GCN stage generations in ONE process (interleaved rounds, HIP events):
plane kernels (stgcn_domain_planes.hip) against the tile kernels (CG_DOM_NO_PLANES=1).  Needs CISTGCN_ABLATION=1 (set here)."""
import os, sys
os.environ["CISTGCN_ABLATION"] = "1"
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


def ab(fn, rounds=5):
    """median over interleaved rounds of (new, old)"""
    new, old = [], []
    for _ in range(rounds):
        os.environ.pop("CG_DOM_NO_PLANES", None)
        new.append(timeit(fn))
        os.environ["CG_DOM_NO_PLANES"] = "1"
        old.append(timeit(fn))
    os.environ.pop("CG_DOM_NO_PLANES", None)
    new.sort(); old.sort()
    return new[len(new) // 2], old[len(old) // 2]


shapes = [(256, 64, 64, 50, 22), (256, 10, 64, 50, 22), (256, 64, 10, 50, 22), (256, 32, 32, 50, 25), (16, 64, 64, 50, 22), (256, 64, 64, 10, 22)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
p = ops._ptr
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
        stats = torch.zeros(2 * co * _lib.STAT_REPLICAS, dtype=torch.float64, device="cuda")
        st = ops._stream(x)
        fwd = lambda: _lib.call("cg_stgcn_domain_fwd", p(x), p(adj), p(w), p(b), p(y), p(stats), B, ci, co, T, V, dom, st)
        bwd = lambda: _lib.call("cg_stgcn_domain_bwd", p(x), p(adj), p(w), p(dy), p(dx), p(dadj), p(dw), p(db), p(ws), B, ci, co, T, V, dom, 0, st)
        # results of both generations
        res = {}
        for name, env in (("new", None), ("old", "1")):
            if env: os.environ["CG_DOM_NO_PLANES"] = env
            else: os.environ.pop("CG_DOM_NO_PLANES", None)
            fwd(); bwd(); torch.cuda.synchronize()
            res[name] = [t.clone() for t in (y, dx, dadj, dw, db)]
        os.environ.pop("CG_DOM_NO_PLANES", None)
        err = [float((a - o).abs().max() / o.abs().max().clamp_min(1e-30)) for a, o in zip(res["new"], res["old"])]
        fn, fo = ab(fwd)
        bn, bo = ab(bwd)
        ng, j = (V, T) if dom == 0 else (T, V)
        bytes_f = 4.0 * (B * ci * T * V + B * ng * j * j + B * co * T * V)
        bytes_b = 4.0 * (2 * B * ci * T * V + 2 * B * ng * j * j + B * co * T * V)
        print("B%d %d->%d T%d V%d dom%d | fwd new %.1f us (%.0f GB/s) old %.1f | bwd new %.1f us (%.0f GB/s) old %.1f | rel diff y %.1e dx %.1e dA %.1e dW %.1e db %.1e" % (
            B, ci, co, T, V, dom, fn, bytes_f / fn / 1e3, fo, bn, bytes_b / bn / 1e3, bo, *err), flush=True)
