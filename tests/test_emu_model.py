"""Whole-model parity on the CPU-only box: the shipped kernels under the test-only HIP shim (tests/hipemu) against the vectors of the
real reference (tests/golden) and against the oracle - eval and train mode, forward, loss, every gradient, attributes, running
statistics, dropout on the masks of the kernels.  The same checks run on the MI355X through libcistgcn_hip.so
(tests/test_gpu_parity.py); full-size parity runs only there."""
import os

import pytest
import torch

import checks
import emu
from helpers import CASES


@pytest.fixture(scope="module", autouse=True)
def _emulated_kernels():
    emu.install()
    yield
    emu.uninstall()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_model_matches_reference_golden(name, mode):
    checks.check_model_golden("cpu", name, modes=(mode,))


@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("fused,staged", [(False, True), (True, False), (False, False)], ids=["generic-domain", "one-launch-per-op", "both"])
def test_model_alternative_launch_plans(mode, fused, staged):
    checks.check_model_golden("cpu", "h36m_c8_t10_v22", modes=(mode,), fused=fused, staged=staged)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_model_stacked_first_level_maps(mode):
    checks.check_model_golden("cpu", "amass_c16_t10_v18", modes=(mode,), stack_all=True)


def test_tiny_model_matches_oracle():
    checks.check_model_vs_oracle("cpu", 4, 4, 5, 2, "eval", To=8, hidden=8, blocks=2, txc=1, stack_all=True)


@pytest.mark.parametrize("cfg", [(8, 10, 22, 8), (64, 10, 22, 6)], ids=str)
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_every_gradient_within_1e4_on_the_same_branches(cfg, mode):
    """north_star tolerance on all parameter gradients with the oracle on the branches the kernels took (see the MI355X test of the same name)"""
    C, T, V, B = cfg
    r = checks.check_model_branch_replay("cpu", C, T, V, B, mode, grad_floor=0.25 if mode == "eval" else 1.0, rel_bound=2e-3)
    print("branch replay %s %s: %s" % (cfg, mode, r))


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_model_without_branch_records(mode):
    """No `act_trace`: the launch plan of production - the first tower level deferred into the collapsing kernels (`tower_maps(defer=True)`),
    which the branch-replay tests switch off because they need the activated maps.  Kink-free network (all slopes 1), every layer stacked."""
    checks.check_model_vs_oracle("cpu", 8, 10, 22, 4, mode, smooth=True, stack_all=True)


def test_tiny_model_every_gradient_strict():
    checks.check_model_branch_replay("cpu", 4, 4, 5, 3, "train", To=8, hidden=8, grad_floor=0.1, blocks=1, txc=1)   # B=3: batch statistics over two samples are ill-conditioned


def test_dropout_step_matches_oracle_on_the_same_masks():
    """train mode WITH dropout 0.1 (the configuration the benchmark times): the oracle applies the keep factors the kernels generated
    at each of its 91 dropout sites and replays the PReLU branches, in fp64"""
    r = checks.check_model_branch_replay("cpu", 8, 10, 22, 8, "train", grad_floor=1.0, max_flip_frac=1e-4, rel_bound=2e-3, oracle_fp64=True, dropout=0.1)
    assert r["dropout_sites"] == 91
    print("dropout replay: %s" % r)


@pytest.mark.parametrize("name", CASES[:2])
def test_golden_case_on_the_branches_of_the_hip_run(name):
    from helpers import load_case, state_of
    rec = load_case(name)
    C, T, V, B = [int(v) for v in rec["meta"]]
    net, ora = checks.build_pair(C, T, V, "cpu", state_of(rec))
    r = checks.check_model_branch_replay("cpu", C, T, V, B, "train", grad_floor=1.0, net=net, ora=ora, x=torch.from_numpy(rec["x"]),
                                         tgt=torch.from_numpy(rec["target"]), oracle_fp64=True, attr_rel=2e-3)
    print("golden %s on the kernels' branches: %s" % (name, r))


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_non_interpretable_layers_at_model_level(mode):
    interp, interp_o = (False, True, False, True, False), (False,)
    checks.check_model_branch_replay("cpu", 8, 10, 22, 6, mode, grad_floor=0.25 if mode == "eval" else 1.0, interp=interp, interp_o=interp_o)


@pytest.mark.skipif(os.environ.get("HIPEMU_SANITIZE", "0") == "1", reason="torch's tracer ends an interpreter that runs under libasan")
def test_model_survives_jit_trace():
    """`writer.add_graph(model, batch)` (train.py:137) = `torch.jit.trace` with its default self-check: the hot path appears as one
    opaque node; in eval mode the self-check passes and the traced module replays bit-identically (train mode with dropout: the
    tracer's re-run draws new masks and warns, as for any dropout model)."""
    import warnings
    net, _ = checks.build_pair(4, 4, 5, "cpu", To=8, hidden=8, dropout=0.1, blocks=1, txc=1)
    net.train()
    x = 50 + 350 * torch.randn(2, 4, 5, 3, generator=torch.Generator().manual_seed(3))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traced = torch.jit.trace(net, x)
    assert any(n.kind() == "prim::PythonOp" for n in traced.graph.nodes())
    net.eval()
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        traced = torch.jit.trace(net, x, check_trace=True)
    bad = [str(w.message) for w in seen if "mismatch" in str(w.message).lower() or "did not match" in str(w.message).lower()]
    assert not bad, bad
    with torch.no_grad():
        assert torch.equal(traced(x)[0], net(x)[0])
