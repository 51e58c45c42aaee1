"""ctypes binding of libcistgcn_hip.so (the C ABI declared in include/cistgcn_hip.h).

There is exactly one backend.  If the library is missing the loader raises; nothing in this
package computes on the CPU or through stock PyTorch operators instead.
"""
import ctypes
import os
from ctypes import POINTER, c_double, c_float, c_int, c_int32, c_longlong, c_uint, c_ulonglong, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcistgcn_hip.so")

_handle = None
STAT_REPLICAS = 16     # CG_STAT_REPLICAS of include/cistgcn_hip.h
ALPHA_SLOTS = 64       # CG_ALPHA_SLOTS: f64 words the partial sums of a shared PReLU slope's gradient are spread over
# Host pointers are refused unless a test harness has injected an emulated build of the very same
# kernel sources (tests/hipemu).  The product never sets this.
_host_pointers_ok = False


class View4(ctypes.Structure):
    _fields_ = [("n", c_longlong * 4), ("s", c_longlong * 4)]


class NormAct(ctypes.Structure):
    _fields_ = [
        ("x", c_void_p), ("xv", View4),
        ("y", c_void_p), ("yv", View4),
        ("pre", c_void_p),
        ("add", c_void_p), ("av", View4),
        ("add_post", c_int),
        ("bn_mode", c_int),
        ("stats", c_void_p),
        ("gamma", c_void_p), ("beta", c_void_p),
        ("running_mean", c_void_p), ("running_var", c_void_p), ("num_batches_tracked", c_void_p),
        ("momentum", c_float), ("eps", c_float),
        ("save_mean", c_void_p), ("save_rstd", c_void_p),
        ("drop_p", c_float), ("seed", c_void_p), ("salt", c_uint),
        ("alpha", c_void_p), ("alpha_n", c_int),
        ("dy", c_void_p), ("dyv", View4),
        ("dx", c_void_p), ("dxv", View4),
        ("dadd", c_void_p), ("dav", View4),
        ("dpre", c_void_p),
        ("red", c_void_p),
        ("dgamma", c_void_p), ("dbeta", c_void_p), ("dalpha", c_void_p),
        ("ystats", c_void_p),
    ]


class ContractDesc(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("X", c_void_p), ("Y", c_void_p), ("bias", c_void_p), ("stats", c_void_p), ("tab", c_void_p),
                ("G", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("splitk", c_int), ("kchunk", c_int),
                ("a_kfast", c_int), ("x_kfast", c_int), ("accumulate", c_int), ("x_vec", c_int), ("stat_ch", c_int), ("mode", c_int), ("block0", c_longlong),
                ("ws", c_void_p), ("chain", c_int), ("pad2", c_int)]


class StatsArgs(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("xv", View4), ("pre", c_void_p), ("stats", c_void_p)]


class SumItem(ctypes.Structure):
    _fields_ = [("a", c_void_p), ("av", View4)]


class Rank1(ctypes.Structure):
    _fields_ = [("s", c_void_p), ("q", c_void_p), ("o", c_void_p), ("dout", c_void_p), ("ds", c_void_p), ("dq", c_void_p),
                ("domain", c_int), ("pad", c_int)]


class CopyItem(ctypes.Structure):
    _fields_ = [("y", c_void_p), ("yv", View4), ("a", c_void_p), ("av", View4)]


class TailBN(ctypes.Structure):
    _fields_ = [("stats", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("num_batches_tracked", c_void_p), ("momentum", c_float), ("eps", c_float), ("save", c_void_p)]


class DstdTail(ctypes.Structure):
    _fields_ = [("B", c_int), ("C", c_int), ("T", c_int), ("V", c_int), ("train", c_int), ("pad0", c_int),
                ("y", c_void_p * 2), ("r", c_void_p * 2), ("w", c_void_p * 2),
                ("bn_t", TailBN * 2), ("alpha_d", c_void_p * 2),
                ("bn_p", TailBN * 2), ("alpha_p", c_void_p * 2),
                ("Wc", c_void_p), ("bn_c", TailBN), ("alpha_c", c_void_p),
                ("gate", c_void_p), ("bres", c_void_p),
                ("drop_p", c_float), ("salt", c_uint * 2), ("pad1", c_int), ("seed", c_void_p),
                ("h0", c_void_p), ("pooled", c_void_p), ("out", c_void_p), ("ostats", c_void_p),
                ("tap_x", c_void_p * 2), ("tap_a", c_void_p * 2), ("tap_h", c_void_p),
                ("dout", c_void_p), ("dpooled", c_void_p), ("dgate", c_void_p), ("red_c", c_void_p),
                ("gp", c_void_p * 2), ("red_p", c_void_p * 2), ("dWc_ws", c_void_p), ("dWc", c_void_p),
                ("dr", c_void_p * 2), ("dw", c_void_p * 2), ("red_t", c_void_p * 2), ("dy", c_void_p * 2),
                ("dgamma_t", c_void_p * 2), ("dbeta_t", c_void_p * 2), ("dalpha_d", c_void_p * 2),
                ("dgamma_p", c_void_p * 2), ("dbeta_p", c_void_p * 2), ("dalpha_p", c_void_p * 2),
                ("dgamma_c", c_void_p), ("dbeta_c", c_void_p), ("dalpha_c", c_void_p)]


class AdjTail(ctypes.Structure):
    _fields_ = [("B", c_int), ("Kc", c_int), ("J", c_int), ("domain", c_int), ("train", c_int), ("pad0", c_int),
                ("s", c_void_p), ("q", c_void_p), ("W0", c_void_p), ("bn", TailBN), ("alpha", c_void_p), ("W4", c_void_p),
                ("drop_p", c_float), ("salt", c_uint), ("seed", c_void_p),
                ("e", c_void_p), ("adj", c_void_p), ("tap", c_void_p),
                ("dadj", c_void_p), ("g", c_void_p), ("red", c_void_p), ("ds", c_void_p), ("dq", c_void_p), ("part", c_void_p),
                ("dW0_ws", c_void_p), ("dW4_ws", c_void_p),
                ("dW0", c_void_p), ("dW4", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p), ("dalpha", c_void_p)]


class PwMaps(ctypes.Structure):
    _fields_ = [("B", c_int), ("Cin", c_int), ("P", c_int), ("n", c_int), ("x", c_void_p),
                ("W", c_void_p * 4), ("M", c_int * 4), ("y", c_void_p * 4), ("stats", c_void_p * 4),
                ("dy", c_void_p * 4), ("dx", c_void_p), ("dW", c_void_p * 4), ("dW_ws", c_void_p),
                ("bias", c_void_p * 4), ("db", c_void_p * 4),
                ("yraw", c_void_p * 4), ("bn_save", c_void_p * 4), ("bn_gamma", c_void_p * 4), ("bn_beta", c_void_p * 4),
                ("bn_red", c_void_p * 4), ("prelu", c_void_p * 4), ("bn_train", c_int), ("pad", c_int)]


class FpnConv(ctypes.Structure):
    _fields_ = [("B", c_int), ("C", c_int), ("O", c_int), ("H", c_int), ("W", c_int), ("n", c_int), ("dil", c_int * 3), ("pad", c_int),
                ("x", c_void_p), ("xs", c_longlong * 3), ("w", c_void_p * 3), ("bias", c_void_p * 3), ("y", c_void_p * 3),
                ("dy", c_void_p * 3), ("dx", c_void_p), ("dw", c_void_p * 3), ("db", c_void_p * 3), ("ws", c_void_p)]


class RowsConv(ctypes.Structure):
    _fields_ = [("B", c_int), ("C", c_int), ("T", c_int), ("V", c_int), ("O", c_int), ("pad", c_int),
                ("x", c_void_p), ("W", c_void_p), ("y", c_void_p), ("stats", c_void_p),
                ("dy", c_void_p), ("dx", c_void_p), ("dW", c_void_p), ("ws", c_void_p),
                ("in_on", c_int), ("in_train", c_int), ("in_bn", TailBN), ("in_alpha", c_void_p), ("in_tap", c_void_p), ("in_red", c_void_p)]


class BlockInput(ctypes.Structure):
    _fields_ = [("B", c_int), ("C", c_int), ("T", c_int), ("V", c_int), ("train", c_int), ("ng", c_int),
                ("x", c_void_p), ("bn", TailBN), ("xn", c_void_p), ("rm", c_void_p), ("rq", c_void_p), ("out", c_void_p),
                ("g", c_void_p * 8), ("dout", c_void_p * 2), ("dout_ld", c_longlong * 2), ("pq", c_void_p), ("gsum", c_void_p), ("red", c_void_p),
                ("dx", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p)]


class GatePath(ctypes.Structure):
    _fields_ = [("z", c_void_p), ("stats", c_void_p), ("stats_ld", c_longlong),
                ("bn2", TailBN), ("alpha2", c_void_p), ("Wl", c_void_p), ("bn3", TailBN), ("alpha3", c_void_p), ("W2", c_void_p),
                ("salt2", c_uint), ("salt3", c_uint), ("y", c_void_p), ("w", c_void_p), ("tap2", c_void_p), ("tap3", c_void_p),
                ("dw", c_void_p), ("dz", c_void_p), ("dstats", c_void_p), ("dWl", c_void_p), ("dW2", c_void_p),
                ("dgamma2", c_void_p), ("dbeta2", c_void_p), ("dalpha2", c_void_p), ("dgamma3", c_void_p), ("dbeta3", c_void_p), ("dalpha3", c_void_p),
                ("scratch", c_void_p), ("red", c_void_p)]


class GateHead(ctypes.Structure):
    _fields_ = [("B", c_int), ("C", c_int), ("S", c_int), ("train", c_int), ("n", c_int), ("pad", c_int),
                ("drop_p", c_float), ("pad2", c_int), ("seed", c_void_p), ("p", GatePath * 2)]


class CtxHeads(ctypes.Structure):
    _fields_ = [("B", c_int), ("P", c_int), ("C", c_int), ("train", c_int), ("x", c_void_p), ("xstats", c_void_p),
                ("w", c_void_p * 2), ("bn", TailBN * 2), ("alpha", c_void_p * 2),
                ("y", c_void_p * 2), ("arg", c_void_p), ("xsave", c_void_p), ("tap", c_void_p * 2),
                ("dy", c_void_p * 2), ("red", c_void_p), ("dx", c_void_p),
                ("dw", c_void_p * 2), ("dgamma", c_void_p * 2), ("dbeta", c_void_p * 2), ("dalpha", c_void_p * 2)]


P = c_void_p
LL = c_longlong
_SIGNATURES = {
    "cg_contract": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, LL, P],
    "cg_contract_many": [POINTER(ContractDesc), c_int, P],
    "cg_contract_kred_ws_floats": [c_int, c_int, c_int],
    "cg_chan_stats": [P, POINTER(View4), P, P, P],
    "cg_chan_stats_many": [POINTER(StatsArgs), c_int, P],
    "cg_norm_act_fwd_many": [POINTER(NormAct), c_int, P],
    "cg_norm_act_bwd_many": [POINTER(NormAct), POINTER(c_int), c_int, P],
    "cg_norm_act_bwd_reduce_many": [POINTER(NormAct), c_int, P],
    "cg_norm_act_params_many": [POINTER(NormAct), c_int, P],
    "cg_copy_many": [POINTER(CopyItem), c_int, P],
    "cg_chan_sum": [P, POINTER(View4), P, P],
    "cg_norm_act_fwd": [POINTER(NormAct), P],
    "cg_norm_act_bwd": [POINTER(NormAct), c_int, P],
    "cg_reduce_bc": [P, POINTER(View4), c_int, P, P, P],
    "cg_reduce_bc_bwd": [P, P, c_int, P, POINTER(View4), P],
    "cg_sum_many": [P, POINTER(View4), POINTER(SumItem), c_int, P],
    "cg_rank1_adj_fwd": [POINTER(Rank1), c_int, c_int, c_int, c_int, P],
    "cg_rank1_adj_bwd": [POINTER(Rank1), c_int, c_int, c_int, c_int, P],
    "cg_gather_joints": [P, P, P, LL, c_int, c_int, P],
    "cg_eval_scatter_mpjpe": [P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "cg_add3": [P, POINTER(View4), P, POINTER(View4), P, POINTER(View4), P, POINTER(View4), P],
    "cg_zero": [P, LL, P],
    "cg_feature_lift_fwd": [P, P, LL, LL, LL, P],
    "cg_feature_lift_bwd": [P, P, P, LL, LL, LL, P],
    "cg_dstd_stats_fwd": [P, P, c_int, c_int, c_int, c_int, P],
    "cg_dstd_stats_bwd": [P, P, P, c_int, c_int, c_int, c_int, P],
    "cg_se_gate_fwd": [P, P, P, P, c_int, c_int, c_int, P],
    "cg_se_gate_bwd": [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "cg_cumsum": [P, POINTER(View4), P, POINTER(View4), c_int, P],
    "cg_mpjpe_fwd": [P, P, P, LL, P],
    "cg_mpjpe_bwd": [P, P, P, P, LL, P],
    "cg_seed_bump": [P, P],
    "cg_stgcn_domain_fwd": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "cg_stgcn_domain_bwd": [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "cg_stgcn_domain_bwd_ws_floats": [c_int, c_int],
    "cg_stgcn_domain_planes_min_workgroups": [c_longlong],
    "cg_multi_copy": [P, P, P, P, P, c_int, P, c_int, c_float, P],
    "cg_scale": [P, LL, c_float, P],
    "cg_dstd_tail_fwd": [POINTER(DstdTail), c_int, P],
    "cg_dstd_tail_bwd": [POINTER(DstdTail), c_int, P],
    "cg_dstd_tail_ws_floats": [c_int],
    "cg_collapse_rows_fwd": [POINTER(RowsConv), P],
    "cg_collapse_rows_bwd": [POINTER(RowsConv), P],
    "cg_collapse_rows_ws_floats": [c_int, c_int, c_int],
    "cg_collapse_cols_fwd": [POINTER(RowsConv), P],
    "cg_collapse_cols_bwd": [POINTER(RowsConv), P],
    "cg_collapse_cols_supported": [c_int, c_int, c_int, c_int],
    "cg_collapse_cols_ws_floats": [c_int, c_int, c_int],
    "cg_fpn_conv_fwd": [POINTER(FpnConv), P],
    "cg_fpn_conv_bwd": [POINTER(FpnConv), P],
    "cg_fpn_conv_supported": [c_int, c_int, c_int, c_int, c_int],
    "cg_fpn_conv_ws_floats": [c_int, c_int, c_int, c_int],
    "cg_pointwise_maps_fwd": [POINTER(PwMaps), P],
    "cg_pointwise_maps_bwd": [POINTER(PwMaps), P],
    "cg_pointwise_maps_ws_floats": [c_int],
    "cg_block_input_fwd": [POINTER(BlockInput), P],
    "cg_block_input_bwd": [POINTER(BlockInput), P],
    "cg_block_input_supported": [c_int, c_int, c_int, c_int],
    "cg_gate_head_fwd": [POINTER(GateHead), P],
    "cg_gate_head_bwd": [POINTER(GateHead), P],
    "cg_gate_head_supported": [c_int, c_int, c_int],
    "cg_gate_head_scratch_floats": [c_int, c_int, c_int],
    "cg_context_heads_fwd": [POINTER(CtxHeads), P],
    "cg_context_heads_bwd": [POINTER(CtxHeads), P],
    "cg_context_heads_red_doubles": [c_int],
    "cg_map2adj_tail_fwd": [POINTER(AdjTail), c_int, c_int, P],
    "cg_map2adj_tail_bwd": [POINTER(AdjTail), c_int, c_int, P],
    "cg_map2adj_tail_ws_floats": [c_int],
    "cg_map2adj_tail_part_floats": [c_int, c_int, c_int],
    "cg_map2adj_tail_red_doubles": [c_int],
    "cg_augment_sequences": [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "cg_adam_flat": [P, P, P, P, LL, c_float, c_float, c_float, c_float, c_float, c_float, c_float, LL, P],
}
EXPORTS = tuple(sorted(_SIGNATURES))


def declare(handle):
    """Attach argument/return types for every entry point of include/cistgcn_hip.h."""
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(handle, name)     # AttributeError if the library does not export it
        fn.argtypes = argtypes
        fn.restype = c_longlong if name.endswith(("_floats", "_doubles", "_workgroups")) else c_int
    return handle


def lib():
    """The loaded kernel library; raises RuntimeError if it has not been built."""
    global _handle
    if _handle is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "cistgcn_amd: %s is missing. Build it with `python -m cistgcn_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no fallback implementation." % LIB_PATH)
        _handle = declare(ctypes.CDLL(LIB_PATH))
    return _handle


_STATUS = {-1: "bad argument (null pointer / option without its buffer)", -2: "unsupported shape"}


def check(status, name):
    if status != 0:
        what = _STATUS.get(status, "hipError_t %d" % status)
        raise RuntimeError("cistgcn_hip: %s failed: %s" % (name, what))


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
