"""Operator parity on the CPU-only box: the shipped .hip kernels, compiled by g++ against the
test-only HIP shim (tests/hipemu), are checked against stock-PyTorch references."""
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
                                   checks.check_stage_kernels, checks.check_flat_adam, checks.check_zero_pool, checks.check_contract_kred, checks.check_contract_stream, checks.check_rank1_adj, checks.check_eval_harness, checks.check_contract_chain, checks.check_dstd_tail, checks.check_map2adj_tail, checks.check_pointwise_maps, checks.check_collapse_rows, checks.check_context_heads, checks.check_block_input, checks.check_tower_maps, checks.check_gate_head], ids=lambda f: f.__name__)
def test_operator(check):
    if check in (checks.check_contract, checks.check_norm_act, checks.check_contract_kred):
        check("cpu", quick=True)
    elif check is checks.check_dstd_tail:
        check("cpu", shapes=((2, 12, 6, 6),))      # (260, 4, 4, 16) = two tiles per workgroup (tile prefetch of K3) passes too: 16 minutes here, covered on the GPU
    elif check is checks.check_collapse_rows:
        check("cpu", shapes=((3, 6, 4, 7, 5), (2, 10, 6, 25, 20)))
    elif check is checks.check_dilated_convs:
        check("cpu", shapes=((2, 5, 4, 10, 7), (3, 6, 5, 4, 6), (2, 20, 10, 6, 6)))
    elif check is checks.check_pointwise_maps:
        check("cpu", shapes=((3, 10, (5, 5, 5, 5), 7, 8), (2, 20, (10, 33), 6, 6), (2, 10, (20, 16), 4, 6), (2, 32, (10, 10, 10), 3, 4), (2, 1, (20, 20), 5, 6), (2, 12, (16, 5), 3, 14), (2, 100, (25,), 3, 4)))
    elif check is checks.check_tower_maps:
        check("cpu", shapes=((3, 10, (5, 5, 5, 5), 7, 8), (2, 20, (10, 16), 6, 6), (2, 12, (16, 5, 7), 3, 14)))
    elif check is checks.check_map2adj_tail:
        check("cpu", shapes=((2, 6, 17), (2, 40, 6)))      # slab counts 6 / 17 / 40: the 16-, 32- and 64-row tiles; 40: the 33..48 range takes the 64-row tile
    else:
        check("cpu")


def test_stgcn_domain_small():
    checks.check_stgcn_domain("cpu", shapes=((3, 10, 8, 5, 7), (2, 3, 3, 6, 9), (2, 18, 16, 5, 7)))


def test_stgcn_domain_planes():
    """plane kernels of the fused ST-GCN stage (forward both domains, both backward kernels) on small instances of every
    instantiated (T, V) family; the full PLANE_SHAPES list runs on the MI355X"""
    checks.check_stgcn_domain("cpu", shapes=((2, 16, 16, 10, 22), (3, 16, 10, 50, 22), (2, 16, 16, 50, 25), (2, 32, 16, 10, 18)), planes=True)
