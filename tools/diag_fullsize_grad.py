#!/usr/bin/env python3
"""Diagnostic (GPU box): full-size train-mode branch replay against the oracle with the ten gradient tensors closest to (or beyond)
the bound listed instead of an assertion at the first one; optional model attributes to switch off (name=0 ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import checks, helpers
flags = dict(a.split("=") for a in sys.argv[1:])


def listing(named_got, named_ref, what="", rel=1e-4, floor=1.0, **kw):
    rows = []
    for k, ref in named_ref.items():
        got = named_got[k]
        got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
        ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else ref
        ok, err, bound = helpers.tol_ok(got, ref, rel, floor)
        rows.append((err / bound, k, err, float(np.abs(ref).max())))
    rows.sort(reverse=True)
    for r in rows[:10]:
        print("  %.3f of the bound  %-60s err %.3e  max|ref| %.3e" % r)
    return rows[0][:2]


helpers.assert_grads_strict = listing
build = checks.build_pair


def build_pair(*a, **k):
    net, ora = build(*a, **k)
    for name, v in flags.items():
        setattr(net, name, bool(int(v)))
        print("net.%s = %s" % (name, bool(int(v))))
    return net, ora


checks.build_pair = build_pair
r = checks.check_model_branch_replay("cuda", 64, 50, 22, 256, "train", grad_floor=0.25, max_flip_frac=1e-4)
print(r)
