"""Whole-model parity on the CPU-only box at a miniature shape (the model is parametric in
(T_in, T_out, V, C, hidden_dim)): the shipped kernels under the test-only HIP shim vs the oracle,
eval and train mode, forward, loss, every gradient, attributes and running statistics.
Full-size parity runs on the MI355X (tests/test_gpu_parity.py)."""
import pytest

import checks
import emu


@pytest.fixture(scope="module", autouse=True)
def _emulated_kernels():
    emu.install()
    yield
    emu.uninstall()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["eval"])          # train mode: the strict test below
def test_tiny_model_matches_oracle(mode):
    checks.check_model_vs_oracle("cpu", 4, 4, 5, 2, mode, To=8, hidden=8)


@pytest.mark.timeout(900)
def test_tiny_model_every_gradient_strict():
    """all parameter gradients within 1e-4 * max(0.1, max|ref|) with the oracle on the branches the kernels took"""
    checks.check_model_branch_replay("cpu", 4, 4, 5, 2, "train", To=8, hidden=8, grad_floor=0.1)
