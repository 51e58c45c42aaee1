#!/usr/bin/env python3
"""Diagnostic (GPU box): does work on a side stream delay the main stream?  A ~2 ms spin kernel goes on the side stream, then an
event is recorded on the main stream; the event's time since a start event on the main stream says whether the two streams share a
hardware queue (HIP maps its streams onto a few of them)."""
import torch
torch.cuda.init()
x = torch.zeros(1, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(1000000); e1.record(); torch.cuda.synchronize()
spin = int(1000000 * 2.0 / e0.elapsed_time(e1))


def probe(main, side, label):
    torch.cuda.synchronize()
    a, b, c = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.cuda.stream(main):
        a.record()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        torch.cuda._sleep(spin)
        c.record()
    with torch.cuda.stream(main):
        x.add_(1.0)
        b.record()
    torch.cuda.synchronize()
    print("%-46s main event after %.3f ms, side done after %.3f ms" % (label, a.elapsed_time(b), a.elapsed_time(c)))


default = torch.cuda.current_stream()
for i in range(10):
    probe(default, torch.cuda.Stream(), "default stream | pool stream %d" % i)
for i in range(4):
    probe(default, torch.cuda.Stream(priority=-1), "default stream | high-priority stream %d" % i)
m = torch.cuda.Stream()
for i in range(6):
    probe(m, torch.cuda.Stream(), "pool stream | pool stream %d" % i)
for i in range(3):
    probe(m, torch.cuda.Stream(priority=-1), "pool stream | high-priority stream %d" % i)
