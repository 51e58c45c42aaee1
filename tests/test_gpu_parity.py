"""Parity tests proper: the HIP path (through the C ABI of libcistgcn_hip.so) on a real MI355X
against (a) stock-PyTorch CPU references per operator, (b) the golden vectors generated from the
real reference, (c) the CPU oracle on fresh seeded inputs.  Run with `-m gpu`."""
import pytest
import torch

import checks
from helpers import CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _real_library():
    from cistgcn_amd import _lib
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _lib._handle = None
    _lib._host_pointers_ok = False
    _lib.lib()          # raises if libcistgcn_hip.so has not been built
    yield


@pytest.mark.parametrize("check", [checks.check_contract, checks.check_norm_act, checks.check_batched_ops, checks.check_dropout,
                                   checks.check_reduce_and_gate, checks.check_copies, checks.check_dilated_convs,
                                   checks.check_stage_kernels, checks.check_stgcn_domain, checks.check_flat_adam, checks.check_zero_pool, checks.check_contract_kred, checks.check_contract_stream, checks.check_rank1_adj, checks.check_eval_harness, checks.check_contract_chain], ids=lambda f: f.__name__)
def test_operator(check):
    check("cuda")


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_model_matches_reference_golden(name, mode):
    checks.check_model_golden("cuda", name, modes=(mode,))


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("fused,staged", [(False, True), (True, False), (False, False)], ids=["generic-domain", "one-launch-per-op", "both"])
def test_model_alternative_launch_plans(mode, fused, staged):
    # fused=False: graph product + channel mix through the generic contraction instead of the fused kernel;
    # staged=False: one launch per op instead of one launch per stage of a block's parallel branches
    checks.check_model_golden("cuda", "h36m_c8_t10_v22", modes=(mode,), fused=fused, staged=staged)


@pytest.mark.parametrize("cfg", [(64, 10, 22, 8), (32, 50, 25, 4), (16, 10, 18, 6), (64, 50, 22, 4)], ids=str)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_model_matches_oracle_wide(cfg, mode):
    C, T, V, B = cfg
    checks.check_model_vs_oracle("cuda", C, T, V, B, mode)


@pytest.mark.parametrize("cfg", [(8, 10, 22, 8), (64, 10, 22, 8), (32, 50, 25, 4)], ids=str)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_gradients_as_accurate_as_cpu_fp32(cfg, mode):
    # kink-free network (all PReLU slopes 1): HIP gradient error vs an fp64 run must be within 8x the error of the
    # reference's own fp32 CPU arithmetic
    C, T, V, B = cfg
    checks.check_model_vs_oracle("cuda", C, T, V, B, mode, smooth=True)


def test_full_size_batch_is_consistent_with_its_chunks():
    """BASELINE configs[2] size (C=64, B=256, 50->25, V=22): the launch plans that only exist at this size (streaming
    contraction, K-reduction weight gradients, many rows per workgroup) against the small-batch plans the oracle tests pin.
    Size-independent properties in eval mode (running statistics, no dropout): (a) a sample's prediction does not depend on
    its batch; (b) MPJPE is a mean over samples, so the parameter gradients of the batch are the mean of its chunks'."""
    from cistgcn_amd import ops
    C, T, V, B, nchunk = 64, 50, 22, 256, 8
    net, _ = checks.build_pair(C, T, V, "cuda")
    net.eval()
    g = torch.Generator().manual_seed(11)
    x = (50 + 350 * torch.randn(B, T, V, 3, generator=g)).cuda()
    tgt = (x[:, -1:].cpu() + 20 * torch.randn(B, 25, V, 3, generator=g)).cuda()
    net.zero_grad()
    pred, = net(x)
    ops.mpjpe(pred, tgt).backward()
    assert any(p.mode == 1 for p in ops._plans.values()) and any(p.mode == 2 for p in ops._plans.values()), "full-size plans not exercised"
    big = {k: p.grad.clone() for k, p in net.named_parameters()}
    pred = pred.detach()
    acc = {k: torch.zeros_like(v) for k, v in big.items()}
    step = B // nchunk
    for c in range(nchunk):
        sl = slice(c * step, (c + 1) * step)
        net.zero_grad()
        pc, = net(x[sl])
        scale = max(1.0, float(pred[sl].abs().max()))
        assert float((pc.detach() - pred[sl]).abs().max()) <= 1e-4 * scale, "prediction of chunk %d depends on the batch" % c
        ops.mpjpe(pc, tgt[sl]).backward()
        for k, p in net.named_parameters():
            acc[k] += p.grad / nchunk
    for k in big:
        ref, got = acc[k].double(), big[k].double()
        rms = float(ref.pow(2).mean().sqrt())
        err = float((got - ref).pow(2).mean().sqrt())
        assert err <= 2e-3 * rms + 1e-7, "gradient of %s: batch vs mean of chunks rms err %.3e (rms %.3e)" % (k, err, rms)


def test_cpu_tensors_are_refused():
    from cistgcn_amd import ops
    with pytest.raises(RuntimeError):
        ops.feature_lift(torch.zeros(2, 10, 22, 3))


def test_graph_replay_matches_eager(branches=False):
    """fwd+loss+bwd captured in a HIP graph replays to the same numbers (bench.py's step)."""
    from cistgcn_amd import ops
    from cistgcn_amd.runtime import GraphedStep
    net, _ = checks.build_pair(8, 10, 22, "cuda")
    net.train()
    g = torch.Generator().manual_seed(5)
    x = (50 + 350 * torch.randn(4, 10, 22, 3, generator=g)).cuda()
    tgt = (x[:, -1:].cpu() + 20 * torch.randn(4, 25, 22, 3, generator=g)).cuda()
    net.dropout = 0.0
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    step = GraphedStep(net, x, tgt, warmup=2, branches=branches, tries=2)     # two captures, the faster one kept: its own grad buffers
    assert len(step.capture_ms) == 2
    net.load_state_dict(sd)
    loss_g = step.replay().item()
    grads_g = [p.grad.clone() for p in net.parameters()]
    net.load_state_dict(sd)
    net.zero_grad()
    net.branch_streams = False
    pred, = net(x)
    loss = ops.mpjpe(pred, tgt)
    loss.backward()
    assert abs(loss.item() - loss_g) <= 1e-5 * max(1.0, abs(loss.item()))
    for a, p in zip(grads_g, net.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-3, atol=max(1e-5, 1e-4 * float(p.grad.abs().max())))   # atomics reorder sums


def test_graph_replay_survives_allocator_churn():
    """Regression: eager allocations of every size class, filled with NaN and freed between replays, must not change the
    captured step.  With hipMemsetAsync nodes in the graph the zero halos of the dilated convolutions picked the NaN up
    (loss and all 698 gradients NaN after the first churn); every zero fill is a kernel now (cg_zero_fill)."""
    from cistgcn_amd.runtime import GraphedStep
    net, _ = checks.build_pair(8, 10, 22, "cuda")
    net.train()
    net.dropout = 0.0
    g = torch.Generator().manual_seed(5)
    x = (50 + 350 * torch.randn(4, 10, 22, 3, generator=g)).cuda()
    tgt = (x[:, -1:].cpu() + 20 * torch.randn(4, 25, 22, 3, generator=g)).cuda()
    step = GraphedStep(net, x, tgt, warmup=2)
    ref = float(step.replay())
    for _ in range(2):
        hold = []
        for n in (1, 16, 64, 256, 1024, 4096, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
            for _k in range(64 if n <= (1 << 16) else 8):
                hold.append(torch.full((n,), float("nan"), device="cuda"))
        torch.cuda.synchronize()
        del hold
        loss = float(step.replay())
        assert abs(loss - ref) <= 1e-5 * abs(ref), (loss, ref)
        assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters())


def test_eval_forward_graph_matches_eager():
    """eval-mode forward captured in a HIP graph (bench.py's forward-only figure) = eager eval forward, also on new data"""
    from cistgcn_amd.runtime import GraphedForward
    net, _ = checks.build_pair(8, 10, 22, "cuda")
    net.eval()
    g = torch.Generator().manual_seed(6)
    x = (50 + 350 * torch.randn(4, 10, 22, 3, generator=g)).cuda()
    fwd = GraphedForward(net, x)
    x2 = (50 + 350 * torch.randn(4, 10, 22, 3, generator=g)).cuda()
    fwd.x.copy_(x2)
    pred_g = fwd.replay().clone()
    with torch.no_grad():
        pred, = net(x2)
    assert torch.allclose(pred_g, pred, rtol=1e-5, atol=1e-3)


def test_training_steps_with_graph_and_flat_adam():
    """End to end: HIP-graph step + flat gradient gather + one-kernel Adam train the model (loss goes down)."""
    from cistgcn_amd.runtime import FlatAdam, GraphedStep
    net, _ = checks.build_pair(8, 10, 22, "cuda")
    net.train()
    net.dropout = 0.1
    g = torch.Generator().manual_seed(9)
    x = (50 + 350 * torch.randn(16, 10, 22, 3, generator=g)).cuda()
    tgt = (x[:, -1:].cpu() + 20 * torch.randn(16, 25, 22, 3, generator=g)).cuda()
    opt = FlatAdam(net, lr=1e-2, weight_decay=1e-4)          # re-homes the parameters before the graph is captured
    step = GraphedStep(net, x, tgt, warmup=2, flat=opt.grads)
    losses = []
    for _ in range(12):
        losses.append(step.replay().item())                  # forward + loss + backward (graph) + gradient gather
        opt.step(gathered=True)
    assert all(l == l for l in losses), losses
    assert losses[-1] < losses[0] - 0.2 and losses[5] < losses[0], losses      # MPJPE (mm) goes down step after step
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())
