#!/usr/bin/env python3
"""Time every contraction launch of one eager training step (HIP events around each launch) and print the
launches by time with their problem shapes - which einsum specs cost what.  GPU box only."""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from cistgcn_amd import ops, _lib
REP = 10

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cistgcn8_b16_t50_v22")
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda:0")
from cistgcn_amd.models import CISTGCN_0
C, B, T, V = bench.WORKLOADS[a.workload]
torch.manual_seed(0)
model = CISTGCN_0(*bench.make_cfg(C, T, V, 0.1)).to(dev).train()
ops.manual_seed(1234, dev)
x, tgt = [t.to(dev) for t in bench.synth(B, T, V, 0)]
log = []
orig = ops._contract_launch
def timed(builders, device, groups=None, chains=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probe = orig(builders, device, groups, chains)
    arrs = [( _lib.ContractDesc * len(probe[c:c + 16]))(*[r.desc for r in probe[c:c + 16]]) for c in range(0, len(probe), 16)]
    st = ops._stream(probe[0].y)
    e0.record()
    for _ in range(REP):       # timing only: split-K / accumulate outputs of this step become garbage
        for arr in arrs: _lib.call("cg_contract_many", arr, len(arr), st)
    e1.record(); e1.synchronize()
    desc = []
    for r in probe:
        d = r.desc
        desc.append("%s G%d M%d N%d K%d sk%d%s%s%s%s" % (r.tag, d.G, d.M, d.N, d.K, d.splitk, " akf" if d.a_kfast else "", " xkf" if d.x_kfast else "",
                                                    " v4" if d.x_vec else "", " acc" if d.accumulate else ""))
    log.append((e0.elapsed_time(e1) * 1e3 / REP, desc))
    return probe
def step():
    for p in model.parameters(): p.grad = None
    ops.begin_step(dev)
    loss = ops.mpjpe(model(x)[0], tgt); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
ops._contract_launch = timed
step()
ops._contract_launch = orig
tot = sum(t for t, _ in log)
print("%d launches, %.1f us total (kernel time, mean of %d back-to-back relaunches)" % (len(log), tot, REP))
agg = collections.defaultdict(lambda: [0, 0.0])
for t, d in log:
    k = " | ".join(d)
    agg[k][0] += 1; agg[k][1] += t
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print("%8.1f us x%-2d  %s" % (t / n, n, k[:600]))
