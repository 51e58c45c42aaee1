"""Data-parallel path on the CPU: two gloo ranks run the runtime's flat-gradient
gather (the HIP multi-copy kernel through the test shim), the all-reduce-mean and the scatter, and
the result must equal the mean of the per-rank gradients.  Per-replica BatchNorm is the reference
semantics, so this is all the communication the path has."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    import emu
    emu.install()
    from cistgcn_amd.runtime import FlatGrads, allreduce_mean_, shard_weights
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 11), torch.nn.BatchNorm1d(11), torch.nn.Linear(11, 5000), torch.nn.PReLU())
    g = torch.Generator().manual_seed(100 + rank)
    for p in net.parameters():
        p.grad = torch.randn(p.shape, generator=g)
    mine = [p.grad.clone() for p in net.parameters()]
    flat = FlatGrads(net.parameters(), "cpu")
    buf = flat.gather()
    assert buf.numel() == sum(p.numel() for p in net.parameters())
    assert torch.equal(buf, torch.cat([m.flatten() for m in mine]))
    allreduce_mean_(buf)
    flat.scatter()
    w = shard_weights(16 if rank == 0 else 48)      # unequal per-GPU batches (BASELINE config 5)
    # by value (numpy): tensors sent through a Queue travel as shared-memory handles that die with this process
    q.put((rank, [p.grad.numpy().copy() for p in net.parameters()], [m.numpy().copy() for m in mine], w))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_mean():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import build_emu
    build_emu.build()          # compile the shim library ONCE here: the two workers would otherwise race to build it
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, reduced, mine, w = q.get(timeout=240)
        res[rank] = (reduced, mine, w)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import numpy as np
    mean = [(a + b) / 2 for a, b in zip(res[0][1], res[1][1])]
    for r in range(world):
        for got, ref in zip(res[r][0], mean):
            assert np.allclose(got, ref, rtol=0, atol=1e-6)
    assert abs(res[0][2] - 0.5) < 1e-6 and abs(res[1][2] - 1.5) < 1e-6
