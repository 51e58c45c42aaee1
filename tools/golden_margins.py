#!/usr/bin/env python3
"""GPU box: error / bound of every comparison of one golden case (tests/checks.check_model_golden), largest first.
Usage: golden_margins.py <case> <mode>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers, checks
rows = []
orig = helpers.assert_close
def spy(a, ref, what, rel=1e-4, floor=1.0):
    r = torch.as_tensor(ref).double().cpu(); g = torch.as_tensor(a).detach().double().cpu()
    err = float((g - r).abs().max()); bound = rel * max(floor, float(r.abs().max()))
    rows.append((err / bound, what, err, bound))
helpers.assert_close = spy; checks.assert_close = spy
case, mode = sys.argv[1], sys.argv[2]
checks.check_model_golden("cuda", case, modes=(mode,))
for r in sorted(rows, reverse=True)[:8]:
    print("%.3f  %s  err %.3e bound %.3e" % r)
