#!/usr/bin/env python3
"""Which device addresses do the captured launches use, and does any of them point into memory the caching allocator
considers FREE after the capture (a dangling pointer inside the graph)?  Hooks the ctypes call layer during capture and
classifies every pointer with torch.cuda.memory_snapshot().  GPU box."""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import checks
from cistgcn_amd import ops, _lib
from cistgcn_amd.runtime import GraphedStep

net, _ = checks.build_pair(8, 10, 22, "cuda"); net.train(); net.dropout = 0.0
g = torch.Generator().manual_seed(5)
x = (50 + 350 * torch.randn(4, 10, 22, 3, generator=g)).cuda()
tgt = (x[:, -1:].cpu() + 20 * torch.randn(4, 25, 22, 3, generator=g)).cuda()

seen = collections.OrderedDict()          # address -> (call name, where)
recording = [False]
def walk(obj, name, path):
    if isinstance(obj, ctypes.c_void_p):
        if obj.value: seen.setdefault(obj.value, (name, path))
    elif isinstance(obj, int):
        pass
    elif isinstance(obj, ctypes.Array):
        for i, e in enumerate(obj): walk(e, name, "%s[%d]" % (path, i))
    elif isinstance(obj, ctypes.Structure):
        for f, _t in obj._fields_:
            v = getattr(obj, f)
            if isinstance(v, int) and _t in (ctypes.c_void_p,):
                if v: seen.setdefault(v, (name, path + "." + f))
            elif isinstance(v, (ctypes.Structure, ctypes.Array)):
                walk(v, name, path + "." + f)
    elif hasattr(obj, "contents"):
        try: walk(obj.contents, name, path + "*")
        except Exception: pass
orig = _lib.call
def call(name, *args):
    if recording[0]:
        for i, a in enumerate(args):
            if isinstance(a, int) and a > (1 << 32): seen.setdefault(a, (name, "arg%d" % i))
            else: walk(a, name, "arg%d" % i)
    return orig(name, *args)
_lib.call = call
ops._lib.call = call

# capture by hand (same as GraphedStep) with recording on during the capture only
step = GraphedStep.__new__(GraphedStep)
step.model, step.flat = net, None
net.branch_streams = False
ops.step_scratch(x.device, True)
step.x, step.target = x.clone(), tgt.clone()
step.params = [p for p in net.parameters() if p.requires_grad]
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): step._step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
from cistgcn_amd.runtime import _drop_graph_attributes
_drop_graph_attributes(net)
step.graph = torch.cuda.CUDAGraph()
recording[0] = True
with torch.cuda.graph(step.graph):
    step.loss = step._step()
recording[0] = False
torch.cuda.synchronize()
snap = torch.cuda.memory_snapshot()
blocks = []
for seg in snap:
    addr = seg["address"]
    for b in seg["blocks"]:
        blocks.append((addr, addr + b["size"], b["state"], tuple(seg.get("segment_pool_id", (0, 0))), b["size"]))
        addr += b["size"]
blocks.sort()
import bisect
starts = [b[0] for b in blocks]
free_hits = collections.Counter(); unknown = 0
examples = {}
for a, (name, path) in seen.items():
    i = bisect.bisect_right(starts, a) - 1
    if i < 0 or not (blocks[i][0] <= a < blocks[i][1]):
        unknown += 1; continue
    lo, hi, state, pool, size = blocks[i]
    if state != "active_allocated":
        if pool != (0, 0):
            continue                       # freed intermediates inside the graph's private pool stay reserved for it: fine
        key = (name, path, "regular", state, size)
        free_hits[key] += 1
        examples.setdefault(key, hex(a))
print("pointers seen during capture:", len(seen), " not in any torch segment:", unknown)
print("pointers into memory that is NOT allocated after the capture:")
for k, v in free_hits.most_common(40):
    print("  %4d  %s  e.g. %s" % (v, k, examples[k]))
