"""Operator parity on the CPU-only box: the shipped .hip kernels, compiled by g++ against the
test-only HIP shim (tests/hipemu), are checked against stock-PyTorch references - on the shape lists the
MI355X run uses (tests/test_gpu_parity.py), minus the few chip-filling ones."""
import os
import subprocess
import sys

import pytest

import checks
import emu


@pytest.fixture(scope="module", autouse=True)
def _emulated_kernels():
    emu.install()
    yield
    emu.uninstall()


@pytest.mark.parametrize("check", [checks.check_contract, checks.check_norm_act, checks.check_batched_ops, checks.check_dropout,
                                   checks.check_reduce_and_gate, checks.check_copies, checks.check_dilated_convs,
                                   checks.check_stage_kernels, checks.check_flat_adam, checks.check_zero_pool, checks.check_contract_kred, checks.check_contract_stream, checks.check_rank1_adj, checks.check_eval_harness, checks.check_contract_chain, checks.check_dstd_tail, checks.check_map2adj_tail, checks.check_pointwise_maps, checks.check_collapse_rows, checks.check_collapse_cols, checks.check_tower_collapse, checks.check_context_heads, checks.check_block_input, checks.check_tower_maps, checks.check_gate_head], ids=lambda f: f.__name__)
def test_operator(check):
    if check is checks.check_context_heads:
        check("cpu", shapes=((3, 5, 12, 7), (4, 25, 66, 64), (2, 3, 10, 33), (16, 25, 75, 64)))
    elif check is checks.check_gate_head:
        check("cpu", shapes=((5, 8, 10, 2), (37, 64, 102, 2), (4, 3, 46, 2), (20, 10, 22, 1), (40, 32, 102, 2)))
    elif check is checks.check_block_input:
        check("cpu", shapes=((3, 5, 4, 6, 3), (2, 10, 10, 22, 7), (4, 64, 5, 22, 8), (2, 6, 5, 5, 2), (3, 3, 22, 25, 4), (9, 32, 50, 25, 6)))
    else:
        check("cpu")


def test_stgcn_domain_small():
    checks.check_stgcn_domain("cpu", shapes=((3, 10, 8, 5, 7), (2, 3, 3, 6, 9), (2, 18, 16, 5, 7)))


def test_stgcn_domain_planes():
    """plane kernels of the fused ST-GCN stage (forward both domains, both backward kernels) on every instantiated (T, V) family"""
    checks.check_stgcn_domain("cpu", shapes=checks.PLANE_SHAPES, planes=True)


@pytest.mark.timeout(1200)
@pytest.mark.skipif(os.environ.get("HIPEMU_SANITIZE", "0") == "1", reason="this IS the sanitizer run")
def test_kernels_under_address_sanitizer():
    """The same kernel sources built with -fsanitize=address (the GPU pool offers no device sanitizer): the operator tests above and
    a part of the model tests of test_emu_model.py (the golden case at T = 10, the tiny models, the dropout step) in a child
    interpreter under libasan; an out-of-bounds LDS / global access or a stack overflow of a kernel aborts the child.  The whole of
    test_emu_model.py passes under the sanitizer too (four minutes; its torch.jit test excepted, which ends the sanitized interpreter
    inside torch's tracer)."""
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan next to g++")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HIPEMU_SANITIZE="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", CISTGCN_ABLATION="1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_emu_ops.py"), os.path.join(here, "test_emu_model.py"), "-x", "-q",
                          "-m", "not gpu", "-p", "no:cacheprovider", "-n", str(min(6, os.cpu_count() or 1)), "-k", "test_operator or test_stgcn or h36m_c8_t10_v22 or tiny or dropout_step"], env=env, capture_output=True, text=True, cwd=os.path.dirname(here), timeout=1100)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "AddressSanitizer" not in res.stderr
