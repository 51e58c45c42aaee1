#!/usr/bin/env python3
"""Where do the memset / device-copy nodes of a training step come from?  Runs one train step of a tiny model
under the test-only HIP shim on the CPU (the aten-level op sequence does not depend on the device) and prints
(a) the aten copy/fill/clone ops by Python call site, (b) cg_zero calls by call site."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import emu, checks
from cistgcn_amd import _lib, ops

emu.install()
zero_sites = collections.Counter()
_orig = _lib.call
def call(name, *a):
    if name == "cg_zero":
        st = traceback.extract_stack(limit=5)[:-1]
        zero_sites[" <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st))] += 1
    return _orig(name, *a)
_lib.call = call

mine, ref = checks.build_pair(4, 4, 5, "cpu", To=8, hidden=8)
torch.manual_seed(1)
x = torch.randn(2, 4, 5, 3); tgt = torch.randn(2, 8, 5, 3)
mine.train()
def step():
    for p in mine.parameters(): p.grad = None
    ops.begin_step(x.device)
    out = mine(x)[0]
    loss = ops.mpjpe(out, tgt)
    loss.backward()
step()
zero_sites.clear()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::add_", "aten::add", "aten::zeros", "aten::contiguous", "aten::cat", "aten::sum", "aten::mul"):
        cnt[(e.name, str(e.input_shapes))] += 1
for k, v in cnt.most_common(40): print("%4d  %-16s %s" % (v, k[0], k[1]))
print("cg_zero:")
for k, v in zero_sites.most_common(): print("%4d  %s" % (v, k))
