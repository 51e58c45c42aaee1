"""CIST-GCN on MI355X: the `nn.Module` surface of the reference model
(human_motion_prediction/models/CISTGCN/CISTGCN.py:478-597) with the arithmetic on hand-written
gfx950 kernels (cistgcn_amd/csrc, called through cistgcn_amd/ops.py).

What is kept from the reference, because its callers depend on it:
class name `CISTGCN`, ctor `(arch, learn)`, `forward(x: (B,T_in,V,3)) -> (pred: (B,T_out,V,3),)`,
every `state_dict` key / shape, the init (same RNG consumption order, so the same seed gives the
same weights), and the interpretation attributes written by each forward
(`st_gcnns.{i}.{dsgn,tsgn}.Adj`, `.w1`, `.w2`, `context_layer.{joints,displacements,seq_joints,
seq_joints_n,seq_joints_dims}`).

The torch leaf modules below (`nn.Conv2d`, `nn.BatchNorm2d`, ...) are parameter holders only; their
`forward` is never called.  There is no CPU path: tensors must live on the HIP device.
"""
import torch
import torch.nn as nn

from ... import ops
from ..layers.SE import SELayer1d, SELayer2d


class Stage(nn.Module):
    """Numbered parameter slots of one reference `nn.Sequential` (gaps = parameter-free steps)."""

    def __init__(self, **slots):
        super().__init__()
        for k, m in slots.items():
            self.add_module(k.lstrip("s"), m)

    def __getitem__(self, i):
        return self._modules[str(i)]


def _conv(cin, cout, k=1, bias=False, **kw):
    return nn.Conv2d(cin, cout, k, bias=bias, **kw)


def _init_small(root, gain, convs):
    """Reference init passes: Map2Adj (CISTGCN.py:175-181, gain .05, convs too) and the model-level
    one (:559-565, gain .1, Linear + PReLU only)."""
    for m in root.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight, gain=gain)
        if convs and isinstance(m, (nn.Conv2d, nn.Conv1d)):
            nn.init.xavier_normal_(m.weight, gain=gain)
        if isinstance(m, nn.PReLU):
            nn.init.constant_(m.weight, 0.25)


class Map2Adj(nn.Module):
    """Parameters of the interpretability attention map (CISTGCN.py:127-181)."""

    def __init__(self, cin, time_dim, joints_dim, domain):
        super().__init__()
        mid = cin // 2
        def tower(kernel, out):
            return Stage(s0=_conv(cin, mid), s1=nn.BatchNorm2d(mid), s2=nn.PReLU(), s3=_conv(mid, mid, kernel),
                         s4=nn.BatchNorm2d(mid), s6=_conv(mid, out))
        self.time_compress = tower((time_dim, 1), time_dim)
        self.joint_compress = tower((1, joints_dim), joints_dim)
        ch = joints_dim if domain == "space" else time_dim
        self.expansor = Stage(s0=_conv(ch, ch), s1=nn.BatchNorm2d(ch), s3=nn.PReLU(), s4=_conv(ch, ch))
        for part in (self.time_compress, self.joint_compress, self.expansor):
            _init_small(part, 0.05, True)


class GraphWeights(nn.Module):
    """`gcn` slot of a domain layer.  Interpretable layers get their adjacency per sample from
    Map2Adj and own no parameter; otherwise a batch-shared A (CISTGCN.py:104-120)."""

    def __init__(self, time_dim, joints_dim, domain, interpretable):
        super().__init__()
        if not interpretable:
            n, s = (time_dim, joints_dim) if domain == "time" else (joints_dim, time_dim)
            self.A = nn.Parameter(torch.empty(n, s, s).uniform_(-1.0 / s ** 0.5, 1.0 / s ** 0.5))


class DomainLayer(nn.Module):
    """Parameters of one Domain_GCNN_layer (CISTGCN.py:208-257); kernel size is [1,1] in every
    instantiation of the reference (:524,:553), so `tcn.0` is a channel-mixing matrix."""

    def __init__(self, cin, cout, time_dim, joints_dim, domain, interpretable):
        super().__init__()
        self.domain, self.interpretable = domain, bool(interpretable)
        self.gcn = GraphWeights(time_dim, joints_dim, domain, interpretable)
        self.tcn = Stage(s0=_conv(cin, cout, bias=True), s1=nn.BatchNorm2d(cout))
        self.residual = Stage(s0=_conv(cin, cout, bias=True), s1=nn.BatchNorm2d(cout)) if cin != cout else nn.Identity()
        self.map_to_adj = Map2Adj(cin, time_dim, joints_dim, domain) if interpretable else nn.Identity()
        self.prelu = nn.PReLU()


class DSTDBlock(nn.Module):
    """Parameters of one DSTD_GC block (CISTGCN.py:289-358)."""

    def __init__(self, cin, cout, interpretable, time_dim, joints_dim, reduction):
        super().__init__()
        self.dsgn = DomainLayer(cin, cout, time_dim, joints_dim, "space", interpretable)
        self.tsgn = DomainLayer(cin, cout, time_dim, joints_dim, "time", interpretable)
        self.compressor = Stage(s0=_conv(2 * cout, cout), s1=nn.BatchNorm2d(cout), s2=nn.PReLU(),
                                s3=SELayer2d(cout, reduction=reduction))
        self.residual = Stage(s0=_conv(cin, cout, bias=True), s1=nn.BatchNorm2d(cout)) if cin != cout else nn.Identity()
        mid = max(1, cout // 2)
        self.global_norm = nn.BatchNorm2d(cin)
        def gate_conv():
            return Stage(s0=_conv(cin, mid, (time_dim, 1)), s1=nn.BatchNorm2d(mid), s3=nn.PReLU(),
                         s4=_conv(mid, cout, (1, joints_dim)), s5=nn.BatchNorm2d(cout), s7=nn.PReLU())
        def gate_map():
            return Stage(s0=nn.Linear(cout + 2 + 2 * time_dim, cout, bias=False), s1=nn.BatchNorm1d(cout), s3=nn.PReLU(),
                         s4=nn.Linear(cout, cout, bias=False))
        self.conv_s, self.conv_t = gate_conv(), gate_conv()
        self.map_s, self.map_t = gate_map(), gate_map()
        self.prelu1 = Stage(s0=nn.BatchNorm2d(cout), s1=nn.PReLU())
        self.prelu2 = Stage(s0=nn.BatchNorm2d(cout), s1=nn.PReLU())


class FPN(nn.Module):
    """Parameters of one time-extrapolator block (CISTGCN.py:38-72)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        if k != 3:
            raise ValueError("txc_kernel_size %d is not supported by the MI355X path (reference YAMLs use 3)" % k)
        for i, d in enumerate((1, 2, 3)):
            self.add_module("block%d" % (i + 1), Stage(s0=_conv(cin, cout, 3, bias=True, padding=d, dilation=d),
                                                       s1=nn.BatchNorm2d(cout), s3=nn.PReLU()))
        self.compress = _conv(3 * cout + cin, cout, bias=True)


class ContextLayer(nn.Module):
    """Parameters of the context branch (CISTGCN.py:394-461)."""

    def __init__(self, hidden, out_seq, joints, reduction):
        super().__init__()
        def ctx_conv(k):
            return Stage(s0=_conv(1, hidden, k), s1=nn.BatchNorm2d(hidden), s2=nn.PReLU())
        self.context_conv1, self.context_conv2, self.context_conv3 = ctx_conv(1), ctx_conv((out_seq, 1)), ctx_conv(1)
        def head():
            return Stage(s0=nn.Linear(hidden, out_seq, bias=False), s2=nn.PReLU())
        self.map1, self.map2, self.map3 = head(), head(), head()
        self.fmap_s = Stage(s0=nn.Linear(3 * out_seq, joints, bias=False), s1=nn.BatchNorm1d(joints))
        self.fmap_t = Stage(s0=nn.Linear(3 * out_seq, out_seq, bias=False), s1=nn.BatchNorm1d(out_seq))
        self.norm_map = Stage(s0=nn.Conv1d(out_seq, out_seq, 1, bias=False), s1=nn.BatchNorm1d(out_seq), s3=nn.PReLU(),
                              s4=SELayer1d(out_seq, reduction=reduction),
                              s5=nn.Conv1d(out_seq, out_seq, 1, bias=False), s6=nn.BatchNorm1d(out_seq), s8=nn.PReLU())
        self.fconv = Stage(s0=_conv(1, 3), s1=nn.BatchNorm2d(3), s2=nn.PReLU(), s3=_conv(3, 3), s4=nn.BatchNorm2d(3), s5=nn.PReLU())
        self.SE = SELayer2d(out_seq, reduction=reduction)


# ---- linear maps as contractions (weights are viewed, never copied) ------------------------------
# Each helper returns (y, stats): stats are the f64 per-channel sums of y from the contraction epilogue
# when `stats=True` (a train-mode BatchNorm follows), else None.
def _pointwise(x, conv, stats=False):
    """1x1 convolution over the channel axis of a 3-D or 4-D (possibly strided) tensor."""
    w = conv.weight.view(conv.out_channels, conv.in_channels)
    if x.dim() == 4 and x.shape[0] * x.shape[2] * x.shape[3] >= _PWM_MIN_POSITIONS and ops.pointwise_maps_ok(x, [w]):
        # long position axis: the stacked pointwise kernel with one map (x, dy read once each backward; the generic contraction split the
        # weight gradient of a 3-channel map over 265 workgroups with atomics: 41 us for 3.4 MB)
        return ops.pointwise_maps(x, [w], stats, biases=[conv.bias])[0]
    spec = "oc,bchw->bohw" if x.dim() == 4 else "oc,bcv->bov"
    bl = "o" if conv.bias is not None else None
    return ops.contract_stats(spec, w, x, conv.bias, bl) if stats else (ops.contract(spec, w, x, conv.bias, bl), None)


_PWM_MIN_POSITIONS = int(__import__("os").environ.get("CISTGCN_PWM_MIN_POSITIONS", "32768"))     # B * H * W from which single 1x1 maps take the stacked kernel


def _collapse_rows(x, conv, stats=False):
    """(H,1) convolution that spans the whole axis 2: (B,C,H,W) -> (B,O,1,W)."""
    w = conv.weight.view(conv.out_channels, conv.in_channels, x.shape[2])
    y, st = ops.contract_stats("och,bchw->bow", w, x) if stats else (ops.contract("och,bchw->bow", w, x), None)
    return y.view(y.shape[0], y.shape[1], 1, y.shape[2]), st


def _collapse_cols(x, conv, stats=False):
    """(1,W) convolution that spans the whole axis 3: (B,C,H,W) -> (B,O,H,1)."""
    w = conv.weight.view(conv.out_channels, conv.in_channels, x.shape[3])
    y, st = ops.contract_stats("ocw,bchw->boh", w, x) if stats else (ops.contract("ocw,bchw->boh", w, x), None)
    return y.view(y.shape[0], y.shape[1], y.shape[2], 1), st


def _linear(x, lin, stats=False):
    return ops.contract_stats("oi,bi->bo", lin.weight, x) if stats else (ops.contract("oi,bi->bo", lin.weight, x), None)


# ---- the same maps as contraction items for ops.contract_many (one launch per stage) ---------------
def _pw_item(x, conv, stats):
    w = conv.weight.view(conv.out_channels, conv.in_channels)
    spec = "oc,bchw->bohw" if x.dim() == 4 else "oc,bcv->bov"
    return (spec, w, x, conv.bias, "o" if conv.bias is not None else None, "o" if stats else None), None


def _rows_item(x, conv, stats):
    w = conv.weight.view(conv.out_channels, conv.in_channels, x.shape[2])
    return ("och,bchw->bow", w, x, None, None, "o" if stats else None), (lambda y: y.view(y.shape[0], y.shape[1], 1, y.shape[2]))


def _cols_item(x, conv, stats):
    w = conv.weight.view(conv.out_channels, conv.in_channels, x.shape[3])
    return ("ocw,bchw->boh", w, x, None, None, "o" if stats else None), (lambda y: y.view(y.shape[0], y.shape[1], y.shape[2], 1))


def _lin_item(x, lin, stats):
    return ("oi,bi->bo", lin.weight, x, None, None, "o" if stats else None), None


def _run_items(items):
    """items: list of (contraction item, view-fixup) -> list of (y, channel sums)"""
    outs = ops.contract_many([it for it, _ in items])
    return [((post(y) if post is not None else y), st) for (y, st), (_, post) in zip(outs, items)]


class CISTGCN(nn.Module):
    """
    Shape:
        - Input:  (N, T_in, V, 3) float32 poses on the HIP device
        - Output: 1-tuple with (N, T_out, V, 3)
    """

    def __init__(self, arch, learn):
        super().__init__()
        p = arch.model_params
        self.clipping = p.clipping
        self.n_input, self.n_output, self.n_joints = p.input_n, p.output_n, p.joints
        self.n_txcnn_layers = p.n_txcnn_layers
        self.txc_kernel_size = [p.txc_kernel_size] * 2
        self.input_gcn, self.output_gcn = p.input_gcn, p.output_gcn
        self.reduction, self.hidden_dim = p.reduction, p.hidden_dim
        self.dropout = float(learn.dropout)
        self.in_ch = 10
        self.fused_domain = True     # False: graph product and channel mix as two generic contractions
        self.staged = True           # True: same-depth ops of a block's parallel branches share one launch
        # True: everything behind the tcn convolutions of a block as phase kernels (ops.dstd_tail); CISTGCN_FUSED_TAIL=0 is a tuning aid
        self.fused_tail = __import__("os").environ.get("CISTGCN_FUSED_TAIL", "1") != "0"
        self.fused_adj = __import__("os").environ.get("CISTGCN_FUSED_ADJ", "1") != "0"
        self.fused_maps = __import__("os").environ.get("CISTGCN_FUSED_MAPS", "1") != "0"
        self.fused_context = __import__("os").environ.get("CISTGCN_FUSED_CONTEXT", "1") != "0"   # ContextLayer heads 1 / 3 without their activations
        self.fused_input = __import__("os").environ.get("CISTGCN_FUSED_INPUT", "1") != "0"   # global_norm + block statistics (and their backward with the fan-in sum) as one operator
        self.fused_defer = __import__("os").environ.get("CISTGCN_FUSED_DEFER", "1") != "0"    # BatchNorm + PReLU of the first tower level applied by the collapsing kernels on load
        self.fused_cols = __import__("os").environ.get("CISTGCN_FUSED_COLS", "1") != "0"      # (1,V) convolutions of the joint towers through csrc/collapse_rows.hip (cg_collapse_cols_*)
        self.fused_towers = __import__("os").environ.get("CISTGCN_FUSED_TOWERS", "1") != "0"   # first tower level + BatchNorm + PReLU as one operator (backward without the BatchNorm input gradient)
        self.fused_gates = __import__("os").environ.get("CISTGCN_FUSED_GATES", "1") != "0"    # the gate paths behind their (1,V) convolutions as one launch
        self.fused_res_maps = True   # the residual 1x1 convolutions (with bias) of a width-changing block through the stacked kernel too
        self.stack_min_elements = int(__import__("os").environ.get("CISTGCN_STACK_MIN_ELEMENTS", str(1 << 21)))      # block inputs smaller than this keep one contraction per first-level map
        # The reference edits the config lists in place (CISTGCN.py:514-517,548); copies are used here
        # so that one `opt` can build several models.
        widths = [self.in_ch] + list(p.input_gcn.model_complexity) + [self.in_ch]
        widths_o = [3] + list(p.output_gcn.model_complexity)
        T, V, To = self.n_input, self.n_joints, self.n_output

        self.st_gcnns = nn.ModuleList()
        self.txcnns = nn.ModuleList()
        self.se = nn.ModuleList()
        self.in_conv = nn.ModuleList()
        self.context_layer = nn.ModuleList()
        self.trans = nn.ModuleList()
        for i in range(len(widths) - 1):
            self.st_gcnns.append(DSTDBlock(widths[i], widths[i + 1], p.input_gcn.interpretable[i], T, V, self.reduction))
        self.context_layer = ContextLayer(self.hidden_dim, To, V, self.reduction)
        self.txcnns.append(FPN(T, To, p.txc_kernel_size))
        for _ in range(1, self.n_txcnn_layers):
            self.txcnns.append(FPN(To, To, p.txc_kernel_size))
        self.prelus = nn.ModuleList(nn.PReLU() for _ in range(self.n_txcnn_layers))
        self.dim_conversor = Stage(s0=_conv(self.in_ch, 3), s1=nn.BatchNorm2d(3), s2=nn.PReLU(), s3=_conv(3, 3), s4=nn.PReLU(3))
        self.st_gcnns_o = nn.ModuleList()
        for i in range(len(widths_o) - 1):
            # the output block runs on (N,3,V,T_out): "time" axis = joints, "joint" axis = frames (:553)
            self.st_gcnns_o.append(DSTDBlock(widths_o[i], widths_o[i + 1], p.output_gcn.interpretable[i], V, To, self.reduction))
        for part in (self.st_gcnns_o, self.st_gcnns, self.txcnns):
            _init_small(part, 0.1, False)
        self._site = 0
        self.backward_cut = None      # (input block index, fn): the step runtime cuts the autograd graph behind that block
        self.act_trace = None         # dict: PReLU module -> (output, post-activation addend) of the last forward (diagnostics)
        self.drop_trace = None        # dict: module naming a dropout site (the BatchNorm in front of it, else the PReLU behind it) -> site id of the last forward (diagnostics)
        self.branch_streams = False   # True: independent branches of a block run on forked HIP streams (runtime.GraphedStep)
        self._streams, self._next_stream = [], 0

    # ---- fork / join of independent branches -----------------------------------------------------
    def _parallel(self, thunks, inputs):
        """Evaluate independent sub-graphs.  With `branch_streams`, all but the last are issued on side streams forked
        from the current stream and joined before returning (the last one stays on the current stream); captured in a HIP
        graph they become parallel branches, and autograd replays each backward on its forward stream, so the small
        kernels of the two branches overlap in both directions.  Fork and join are the only cross-stream edges: every
        tensor that crosses is produced before the fork or consumed after the join, which is also what keeps the caching
        allocator's per-stream reuse safe without record_stream."""
        if not (self.branch_streams and inputs[0].is_cuda) or len(thunks) < 2:
            return [t() for t in thunks]
        cur = torch.cuda.current_stream()
        while len(self._streams) < len(thunks) - 1:
            self._streams.append(torch.cuda.Stream(device=inputs[0].device))
        results = []
        for t, s in zip(thunks[:-1], self._streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                results.append(t())
        results.append(thunks[-1]())
        for s in self._streams[:len(thunks) - 1]:
            cur.wait_stream(s)
        return results

    # ---- fused row op with per-call dropout site id -------------------------------------------
    def _na(self, x, bn=None, prelu=None, drop=False, **kw):
        """x is a tensor or the (y, channel-sums) pair returned by the linear-map helpers."""
        return self._na_many([dict(x=x, bn=bn, prelu=prelu, drop=drop, **kw)])[0]

    def _lin(self, fn, x, layer, bn=True):
        """linear map `fn` followed (bn=True) by a BatchNorm: ask the kernel for channel sums in train mode"""
        return fn(x, layer, stats=bn and self.training)

    # ---- Map2Adj.forward, CISTGCN.py:183-189 ----------------------------------------------------
    def _adjacency(self, layer, x):
        m = layer.map_to_adj
        B, _, T, V = x.shape

        def tower(t, collapse):
            h = self._na(self._lin(_pointwise, x, t[0]), bn=t[1], prelu=t[2])
            h = self._na(self._lin(collapse, h, t[3]), bn=t[4], drop=True)
            return _pointwise(h, t[6])[0]

        q, s = self._parallel([lambda: tower(m.time_compress, _collapse_rows), lambda: tower(m.joint_compress, _collapse_cols)], [x])
        q, s = q.view(B, T, V), s.view(B, V, T)                       # q[b,tau,v], s[b,v,t]
        if layer.domain == "space":
            o = ops.contract("bvt,bxv->bvtx", s, q)                  # o[b,v,t,tau] = s[b,v,t] q[b,tau,v]
        else:
            o = ops.contract("bvt,btw->btvw", s, q)                  # o[b,t,v,w]  = s[b,v,t] q[b,t,w]
        e = m.expansor
        h = self._na(self._lin(_pointwise, o, e[0]), bn=e[1], drop=True, prelu=e[3])
        return _pointwise(h, e[4])[0]

    def _fused_stage_ok(self, conv):
        """The fused graph-product + channel-mix kernel takes layers up to 128 channels (its LDS image); wider layers (any
        `model_complexity` is legal in the reference) go through the two generic contractions."""
        return self.fused_domain and conv.in_channels <= 128 and conv.out_channels <= 128

    # ---- Domain_GCNN_layer.forward, CISTGCN.py:259-269 -------------------------------------------
    def _domain(self, layer, xn):
        res = xn if isinstance(layer.residual, nn.Identity) else self._na(self._lin(_pointwise, xn, layer.residual[0]), bn=layer.residual[1])
        conv = layer.tcn[0]
        if layer.interpretable:
            adj = self._adjacency(layer, xn)
            layer.Adj = adj
            if self._fused_stage_ok(conv):
                w = conv.weight.view(conv.out_channels, conv.in_channels)
                y = ops.stgcn_domain(xn, adj, w, conv.bias, 0 if layer.domain == "space" else 1, self.training)
            else:
                spec = "bctv,bvtq->bcqv" if layer.domain == "space" else "bctv,btvw->bctw"
                y = self._lin(_pointwise, ops.contract(spec, xn, adj), conv)
        else:
            layer.Adj = xn
            spec = "bctv,vtq->bcqv" if layer.domain == "space" else "bctv,tvw->bctw"
            y = self._lin(_pointwise, ops.contract(spec, xn, layer.gcn.A), conv)
        return self._na(y, bn=layer.tcn[1], drop=True, add=res, prelu=layer.prelu)

    # ---- gate path of DSTD_GC.forward, CISTGCN.py:378-384 ----------------------------------------
    def _gate(self, conv, mp, xn, stats):
        h = self._na(self._lin(_collapse_rows, xn, conv[0]), bn=conv[1], drop=True, prelu=conv[3])
        h = self._na(self._lin(_collapse_cols, h, conv[4]), bn=conv[5], drop=True, prelu=conv[7])
        h = ops.cat_channels([h.view(h.shape[0], -1), stats])
        h = self._na(self._lin(_linear, h, mp[0]), bn=mp[1], drop=True, prelu=mp[3])
        return _linear(h, mp[4])[0]

    # ---- DSTD_GC.forward, CISTGCN.py:373-390 ------------------------------------------------------
    def _block(self, m, x):
        xn = self._na(x, bn=m.global_norm)
        stats = ops.dstd_stats(xn)                      # computed once; the reference evaluates it twice (:377,:379)
        m.w1, m.w2, x1, x2 = self._parallel([lambda: self._gate(m.conv_s, m.map_s, xn, stats),
                                             lambda: self._gate(m.conv_t, m.map_t, xn, stats),
                                             lambda: self._domain(m.dsgn, xn), lambda: self._domain(m.tsgn, xn)], [xn, stats])
        a = self._na(x1, pre=m.w1, bn=m.prelu1[0], prelu=m.prelu1[1])
        b = self._na(x2, pre=m.w2, bn=m.prelu2[0], prelu=m.prelu2[1])
        c = m.compressor
        h = self._na(self._lin(_pointwise, ops.cat_channels([a, b]), c[0]), bn=c[1], prelu=c[2])
        gate = ops.se_gate(ops.mean_bc(h), c[3].w1, c[3].w2)
        res = xn if isinstance(m.residual, nn.Identity) else self._na(self._lin(_pointwise, xn, m.residual[0]), bn=m.residual[1])
        return self._na(h, pre=gate, add=res, add_post=True)

    # ---- the same block with horizontal fusion ---------------------------------------------------------
    def _na_many(self, calls):
        """calls: dicts for ops.norm_act (x may be a (tensor, sums) pair); adds train flag, dropout and site ids."""
        out = []
        for kw in calls:
            kw = dict(kw)
            self._site += 1
            drop = kw.pop("drop", False)
            kw.update(train=self.training, drop_p=self.dropout if drop else 0.0, salt=self._site)
            if drop and self.drop_trace is not None:
                self.drop_trace[kw.get("bn") if kw.get("bn") is not None else kw.get("prelu")] = self._site
            out.append(kw)
        ys = ops.norm_act_many(out)
        if self.act_trace is not None:
            # branch record of every PReLU (its output, and the addend when that is added behind the activation):
            # lets a checker replay another implementation of the network on the very same branches
            for kw, y in zip(out, ys):
                if kw.get("prelu") is not None:
                    y = y[0] if isinstance(y, tuple) else y
                    add = kw.get("add") if kw.get("add_post") else None
                    self.act_trace[kw["prelu"]] = (y.detach(), add.detach() if add is not None else None)
        return ys

    @staticmethod
    def _map_groups(x, ws):
        """index groups of the 1x1 maps `ws` of x that `ops.pointwise_maps` takes in one launch each (at most four maps and 128
        stacked rows, every map rounded up to 16 rows); None when one of them does not fit the kernel at all"""
        groups, rows = [], 0
        for i, w in enumerate(ws):
            if not ops.pointwise_maps_ok(x, [w]):
                return None
            r = (w.shape[0] + 15) // 16 * 16
            if not groups or rows + r > 128 or len(groups[-1]) == 4:
                groups.append([])
                rows = 0
            groups[-1].append(i)
            rows += r
        return groups

    def _block_staged(self, m, x):
        """DSTD_GC.forward (CISTGCN.py:373-390) with the gate paths (:378-384), the four Map2Adj towers (:183-189) and
        the two domain layers (:259-269) advanced in lock-step: every stage is one contraction launch or one row-kernel
        launch for all branches instead of one launch per branch and op."""
        tr = self.training
        doms = (m.dsgn, m.tsgn)
        if not all(d.interpretable for d in doms):
            return self._block(m, x)
        x_in, x_sums = x if isinstance(x, tuple) else (x, None)       # (tensor, f64 channel sums) from the previous block's tail in train mode
        fused_in = self.fused_input and ops.block_input_ok(x_in)
        xn0 = None if fused_in else self._na(x, bn=m.global_norm)
        B, _, T, V = x_in.shape
        has_res = not isinstance(m.dsgn.residual, nn.Identity)
        has_bres = not isinstance(m.residual, nn.Identity)
        # the normalised input feeds the statistics, the first-level maps, both graph stages and the identity residuals:
        # one alias per consumer, so that backward sums their gradients in one launch (ops.fanout)
        maps = [d.map_to_adj for d in doms]
        tower_in = [c for a in maps for c in (a.time_compress[0], a.joint_compress[0])]
        tower_w = [c.weight.view(c.out_channels, c.in_channels) for c in tower_in]
        # the four tower convolutions read the block input in one pass, forward and backward (csrc/tower_maps.hip)
        # (both stackings below pay off once the block input is worth the two or three extra small launches: measured break-even
        # between the B=16 / C=8 and the B=256 / C=64 workloads)
        big = x_in.numel() >= self.stack_min_elements
        stacked = big and self.fused_maps and all(c.bias is None for c in tower_in) and ops.pointwise_maps_ok(x_in, tower_w)
        cs, ct = m.conv_s[0], m.conv_t[0]
        gates = big and self.fused_maps and cs.weight.shape == ct.weight.shape and cs.bias is None and ct.bias is None
        rows_gate = gates and 2 * cs.out_channels <= 64 and ops.collapse_rows_ok(x_in, cs.weight.view(cs.out_channels, cs.in_channels, -1))
        # the residual maps of a block that changes its width (1x1 convolutions WITH bias, CISTGCN.py:246-254 / :357-365) read the block
        # input in one pass per group as well (a group: up to 128 stacked output rows)
        res_convs = ([d.residual[0] for d in doms] if has_res else []) + ([m.residual[0]] if has_bres else [])
        res_w = [c.weight.view(c.out_channels, c.in_channels) for c in res_convs]
        res_groups = self._map_groups(x_in, res_w) if (big and self.fused_maps and self.fused_res_maps and res_convs) else None
        n_alias = 4 + (1 if stacked else 0) + (1 if rows_gate else 0) + (len(res_groups) if res_groups else 0) + (0 if has_res else 2) + (0 if has_bres else 1)
        if fused_in:
            # global_norm and the block statistics from one pass over the block input; in backward the gradients of all the consumers below,
            # the statistics' gradient and the BatchNorm backward are two streaming passes (csrc/block_input.hip)
            self._site += 1          # the row kernel of global_norm numbers a (dropout-free) site: later layers draw the same masks on either path
            xa, (stats_s, stats_t) = ops.block_input(x_in, m.global_norm, tr, n_alias - 1, stats=x_sums if tr else None)
            xa = [None] + xa
        else:
            xa = list(ops.fanout(xn0, n_alias))
        x_stats, xn, x_dom = xa[0], xa[1], xa[2:4]
        k = 4
        x_maps = x_gates = None
        if stacked:
            x_maps, k = xa[4], 5
        if rows_gate:
            x_gates, k = xa[k], k + 1
        x_resmaps = []
        if res_groups:
            x_resmaps, k = xa[k:k + len(res_groups)], k + len(res_groups)
        x_res = xa[k:k + 2] if not has_res else None
        x_bres = xa[-1] if not has_bres else None
        if not fused_in:
            stats_s, stats_t = ops.fanout(ops.dstd_stats(x_stats), 2)    # one alias per gate path
        # 1. every first-level map of xn
        # the two gate paths start with the same (T,1) convolution shape on the same input: one convolution of the stacked weights
        # (the block input travels once instead of twice, forward and in both gradients)
        gate_rows = None
        if gates:
            O = cs.out_channels
            w2 = ops.cat_channels([cs.weight.view(1, O, -1), ct.weight.view(1, O, -1)]).view(2 * O, cs.in_channels, xn.shape[2])
            if x_gates is not None:
                gate_rows = ops.collapse_rows(x_gates, w2)[0]            # whole-sample kernel (csrc/collapse_rows.hip)
                items = []
            else:
                items = [(("och,bchw->bow", w2, xn, None, None, None), None)]
        else:
            items = [_rows_item(xn, cs, tr), _rows_item(xn, ct, tr)]
        if not stacked:
            for a in maps:
                items += [_pw_item(xn, a.time_compress[0], tr), _pw_item(xn, a.joint_compress[0], tr)]
        if res_groups is None:
            items += [_pw_item(xn, c, tr) for c in res_convs]
        o = _run_items(items) if items else []
        if gates:
            yg = gate_rows if gate_rows is not None else o.pop(0)[0]
            o = [(g.unsqueeze(2), None) for g in ops.split_channels(yg, (O, O))] + o
        towers_done, tower_tr = None, None
        if stacked:
            tb = [b for a in maps for b in (a.time_compress[1], a.joint_compress[1])]
            tp = [b for a in maps for b in (a.time_compress[2], a.joint_compress[2])]
            if self.fused_towers and ops.tower_maps_ok(x_maps, tower_w, tp):
                # first level of the four towers with its BatchNorm + PReLU as one operator: backward never stores the gradient in front of
                # the BatchNorm (the pointwise backward undoes BatchNorm and PReLU while loading)
                # ... and, when every map goes into a whole-sample collapsing kernel, the BatchNorm + PReLU themselves move into that
                # kernel's load path: the four activated (B, C/2, T, V) maps are never stored (unless branch records are asked for: `in_tap`)
                def _collapse_ok(k, wmap):
                    a_, tower = maps[k // 2], (maps[k // 2].time_compress if k % 2 == 0 else maps[k // 2].joint_compress)
                    c_ = tower[3]
                    probe = torch.empty(0, device=x_maps.device).new_empty((B, wmap.shape[0], T, V))
                    wv = c_.weight.view(c_.out_channels, c_.in_channels, -1)
                    if c_.bias is not None:
                        return False
                    return ops.collapse_rows_ok(probe, wv) if k % 2 == 0 else (self.fused_cols and ops.collapse_cols_ok(probe, wv))
                defer = self.fused_defer and big and all(_collapse_ok(k, wmap) for k, wmap in enumerate(tower_w))
                if defer:
                    towers_done, tower_tr = ops.tower_maps(x_maps, tower_w, tb, tp, tr, defer=True)
                    if self.act_trace is not None:                       # branch records: the collapsing kernels write the activated maps out as well
                        for tr_ in tower_tr:
                            tr_["want_tap"] = True
                else:
                    towers_done = ops.tower_maps(x_maps, tower_w, tb, tp, tr)
                    if self.act_trace is not None:
                        for mod, h in zip(tp, towers_done):
                            self.act_trace[mod] = (h.detach(), None)
                o = o[:2] + [None] * 4 + o[2:]
            else:
                o = o[:2] + ops.pointwise_maps(x_maps, tower_w, tr) + o[2:]
        res_done = None
        if res_groups:
            res_bns = ([d.residual[1] for d in doms] if has_res else []) + ([m.residual[1]] if has_bres else [])
            if self.fused_towers and all(ops.tower_maps_ok(xg, [res_w[i] for i in grp], [None] * len(grp)) for xg, grp in zip(x_resmaps, res_groups)):
                # the residual maps with their BatchNorm as one operator per group, like the towers: backward is one reduction pass and the
                # pointwise backward undoing the BatchNorm on load (as two operators: 229 us of cg_norm_act_bwd on the 10 -> 64 block)
                res_done = [None] * len(res_convs)
                for xg, grp in zip(x_resmaps, res_groups):
                    hs = ops.tower_maps(xg, [res_w[i] for i in grp], [res_bns[i] for i in grp], [None] * len(grp), tr, biases=[res_convs[i].bias for i in grp])
                    for i, h in zip(grp, hs):
                        res_done[i] = h
                o = o + [None] * len(res_convs)
            else:
                ro = [None] * len(res_convs)
                for xg, grp in zip(x_resmaps, res_groups):
                    for i, y in zip(grp, ops.pointwise_maps(xg, [res_w[i] for i in grp], tr, biases=[res_convs[i].bias for i in grp])):
                        ro[i] = y
                o = o + ro
        gs, gt, tc = o[0], o[1], o[2:6]
        # 2. their BatchNorm / PReLU tails
        calls = [dict(x=gs, bn=m.conv_s[1], drop=True, prelu=m.conv_s[3]), dict(x=gt, bn=m.conv_t[1], drop=True, prelu=m.conv_t[3])]
        if towers_done is None:
            for i, a in enumerate(maps):
                calls += [dict(x=tc[2 * i], bn=a.time_compress[1], prelu=a.time_compress[2]),
                          dict(x=tc[2 * i + 1], bn=a.joint_compress[1], prelu=a.joint_compress[2])]
        k = 6
        n_tower_calls = len(calls) - 2
        if res_done is None:
            if has_res:
                calls += [dict(x=o[k + i], bn=d.residual[1]) for i, d in enumerate(doms)]
                k += 2
            if has_bres:
                calls.append(dict(x=o[k], bn=m.residual[1]))
        # site numbers as on the row-kernel path in every case: gates, four (dropout-free) tower sites, residual maps
        if towers_done is None:
            r = self._na_many(calls)
        else:
            r = self._na_many(calls[:2])
            self._site += 4
            r = r + towers_done + (self._na_many(calls[2 + n_tower_calls:]) if len(calls) > 2 + n_tower_calls else [])
        if res_done is not None:
            self._site += len(res_done)
            r = r + res_done
        gs, gt, t1 = r[0], r[1], r[2:6]
        res = r[6:8] if has_res else x_res
        bres = r[-1] if has_bres else x_bres
        # 3. collapsing convolutions
        items = [_cols_item(gs, m.conv_s[4], tr), _cols_item(gt, m.conv_t[4], tr)]
        rows3, cols3 = {}, {}
        for i, a in enumerate(maps):
            c3 = a.time_compress[3]
            x3 = t1[2 * i][0] if isinstance(t1[2 * i], tuple) else t1[2 * i]
            w3 = c3.weight.view(c3.out_channels, c3.in_channels, -1)
            if big and c3.bias is None and ops.collapse_rows_ok(x3, w3):
                y3, st3 = ops.collapse_rows(x3, w3, tr, transform=tower_tr[2 * i] if tower_tr else None)      # whole-sample kernel (csrc/collapse_rows.hip)
                rows3[i] = (y3.unsqueeze(2), st3)
            else:
                assert tower_tr is None, "a deferred tower map needs its collapsing kernel"
                items.append(_rows_item(t1[2 * i], c3, tr))
            c4 = a.joint_compress[3]
            x4 = t1[2 * i + 1][0] if isinstance(t1[2 * i + 1], tuple) else t1[2 * i + 1]
            w4 = c4.weight.view(c4.out_channels, c4.in_channels, -1)
            if big and self.fused_cols and c4.bias is None and ops.collapse_cols_ok(x4, w4):
                y4, st4 = ops.collapse_cols(x4, w4, tr, transform=tower_tr[2 * i + 1] if tower_tr else None)  # whole-sample kernel for the joint axis
                cols3[i] = (y4.unsqueeze(3), st4)
            else:
                assert tower_tr is None, "a deferred tower map needs its collapsing kernel"
                items.append(_cols_item(t1[2 * i + 1], c4, tr))
        if tower_tr and self.act_trace is not None:
            tp_ = [b_ for a_ in maps for b_ in (a_.time_compress[2], a_.joint_compress[2])]
            for mod, tr_ in zip(tp_, tower_tr):
                self.act_trace[mod] = (tr_["tap"].detach(), None)
        o = _run_items(items)
        if rows3 or cols3:                                               # back into the order gates | (time, joint) per tower
            rest, o = o[2:], o[:2]
            for i in range(len(maps)):
                o.append(rows3[i] if i in rows3 else rest.pop(0))
                o.append(cols3[i] if i in cols3 else rest.pop(0))
        # 4. BatchNorm tails
        Cg = m.conv_s[4].out_channels
        gate_fused = (self.fused_gates and o[0][0].shape[1] == Cg and o[0][0].numel() == B * Cg and
                      ops.gate_head_ok(B, Cg, stats_s.shape[1], (m.conv_s[7], m.conv_t[7], m.map_s[3], m.map_t[3])))
        tower_calls = []
        for i, a in enumerate(maps):
            tower_calls += [dict(x=o[2 + 2 * i], bn=a.time_compress[4], drop=True), dict(x=o[3 + 2 * i], bn=a.joint_compress[4], drop=True)]
        if gate_fused:
            # everything of the two gate paths behind their (1,V) convolutions in ONE launch (csrc/gate_head.hip); the site ids are the ones
            # the row-kernel chain numbers: conv_s.5 / conv_t.5 in front of the four tower sites, map_s.1 / map_t.1 behind them
            s2 = (self._site + 1, self._site + 2)
            self._site += 2
            r = [None, None] + self._na_many(tower_calls)
            s3 = (self._site + 1, self._site + 2)
            self._site += 2
            if self.drop_trace is not None:
                self.drop_trace[m.conv_s[5]], self.drop_trace[m.conv_t[5]], self.drop_trace[m.map_s[1]], self.drop_trace[m.map_t[1]] = s2[0], s2[1], s3[0], s3[1]
            taps = [] if self.act_trace is not None else None
            m.w1, m.w2 = ops.gate_head([o[0][0].view(B, Cg), o[1][0].view(B, Cg)], [stats_s, stats_t], [m.conv_s, m.conv_t], [m.map_s, m.map_t], tr,
                                       drop_p=self.dropout, salts=((s2[0], s3[0]), (s2[1], s3[1])), taps=taps)
            if taps is not None:
                self.act_trace[m.conv_s[7]], self.act_trace[m.map_s[3]] = (taps[0], None), (taps[1], None)
                self.act_trace[m.conv_t[7]], self.act_trace[m.map_t[3]] = (taps[2], None), (taps[3], None)
            items, g = [], None
        else:
            calls = [dict(x=o[0], bn=m.conv_s[5], drop=True, prelu=m.conv_s[7]), dict(x=o[1], bn=m.conv_t[5], drop=True, prelu=m.conv_t[7])]
            r = self._na_many(calls + tower_calls)
            hs = ops.cat_channels([r[0].view(B, -1), stats_s])
            ht = ops.cat_channels([r[1].view(B, -1), stats_t])
            items = [_lin_item(hs, m.map_s[0], tr), _lin_item(ht, m.map_t[0], tr)]
        # 5. gate Linear + last tower maps
        for i, a in enumerate(maps):
            items += [_pw_item(r[2 + 2 * i], a.time_compress[6], False), _pw_item(r[3 + 2 * i], a.joint_compress[6], False)]
        o = _run_items(items)
        if gate_fused:
            o = [None, None] + o
        else:
            g = self._na_many([dict(x=o[0], bn=m.map_s[1], drop=True, prelu=m.map_s[3]), dict(x=o[1], bn=m.map_t[1], drop=True, prelu=m.map_t[3])])
        # 6. rank-1 products  o[b,v,t,tau] = s[b,v,t] q[b,tau,v]  |  o[b,t,v,w] = s[b,v,t] q[b,t,w]
        seeds = []
        for i, d in enumerate(doms):
            q, s = o[2 + 2 * i][0].view(B, T, V), o[3 + 2 * i][0].view(B, V, T)
            seeds.append((0 if d.domain == "space" else 1, s, q))
        if self.fused_adj and T <= 64 and V <= 64:
            # seed, expansor and adjacency of both towers in two phase launches (csrc/map2adj_tail.hip); the seed is never stored
            if not gate_fused:
                o = _run_items([_lin_item(g[0], m.map_s[4], False), _lin_item(g[1], m.map_t[4], False)])
                m.w1, m.w2 = o[0][0], o[1][0]
            self._site += 2
            if self.drop_trace is not None:
                self.drop_trace[maps[0].expansor[1]], self.drop_trace[maps[1].expansor[1]] = self._site - 1, self._site
            taps = [] if self.act_trace is not None else None
            adj = ops.map2adj_tail(seeds, [a.expansor for a in maps], tr, drop_p=self.dropout, salts=(self._site - 1, self._site), taps=taps)
            adj = [(a, None) for a in adj]
            if taps is not None:
                for a, tap in zip(maps, taps):
                    self.act_trace[a.expansor[3]] = (tap, None)
        else:
            if T <= 64 and V <= 64:
                oo = ops.rank1_adj(seeds)                        # both towers in one launch; backward reads each d o once
            else:
                oo = [y for y, _ in ops.contract_many([("bvt,bxv->bvtx" if dom == 0 else "bvt,btw->btvw", s, q, None, None, None)
                                                       for dom, s, q in seeds])]
            # 7. gate output Linear + expansor first map
            items = [] if gate_fused else [_lin_item(g[0], m.map_s[4], False), _lin_item(g[1], m.map_t[4], False)]
            items += [_pw_item(oo[i], a.expansor[0], tr) for i, a in enumerate(maps)]
            o = _run_items(items)
            if gate_fused:
                o = [None, None] + o
            else:
                m.w1, m.w2 = o[0][0], o[1][0]
            e = self._na_many([dict(x=o[2 + i], bn=a.expansor[1], drop=True, prelu=a.expansor[3]) for i, a in enumerate(maps)])
            adj = _run_items([_pw_item(e[i], a.expansor[4], False) for i, a in enumerate(maps)])
        # 8. graph product + channel mix, then BN + residual + PReLU
        ys = []
        for i, d in enumerate(doms):
            d.Adj = adj[i][0]
            conv = d.tcn[0]
            if self._fused_stage_ok(conv):
                wmat = conv.weight.view(conv.out_channels, conv.in_channels)
                ys.append(ops.stgcn_domain(x_dom[i], d.Adj, wmat, conv.bias, 0 if d.domain == "space" else 1, tr))
            else:
                spec = "bctv,bvtq->bcqv" if d.domain == "space" else "bctv,btvw->bctw"
                ys.append(self._lin(_pointwise, ops.contract(spec, x_dom[i], d.Adj), conv))
        c = m.compressor
        if self.fused_tail and c[0].out_channels <= 64 and all(isinstance(y, tuple) and (y[1] is not None or not tr) for y in ys):
            # everything behind the two tcn convolutions in five phase launches (csrc/dstd_tail.hip)
            self._site += 2
            if self.drop_trace is not None:
                self.drop_trace[doms[0].tcn[1]], self.drop_trace[doms[1].tcn[1]] = self._site - 1, self._site
            taps = [] if self.act_trace is not None else None
            out, ost = ops.dstd_tail([y[0] for y in ys], [y[1] for y in ys], res, (m.w1, m.w2),
                                     (doms[0].tcn[1], doms[1].tcn[1], m.prelu1[0], m.prelu2[0], c[1]),
                                     (doms[0].prelu, doms[1].prelu, m.prelu1[1], m.prelu2[1], c[2]), c[0].weight, c[3], bres, tr,
                                     drop_p=self.dropout, salts=(self._site - 1, self._site), emit_stats=tr, taps=taps)
            if taps is not None:
                for mod, tap in zip((doms[0].prelu, doms[1].prelu, m.prelu1[1], m.prelu2[1], c[2]), taps):
                    self.act_trace[mod] = (tap, None)
            self._site += 4          # the row-kernel chain below numbers four more (dropout-free) sites: later blocks draw the same masks on either path
            return (out, ost) if tr else out
        x12 = self._na_many([dict(x=ys[i], bn=d.tcn[1], drop=True, add=res[i], prelu=d.prelu) for i, d in enumerate(doms)])
        ab = self._na_many([dict(x=x12[0], pre=m.w1, bn=m.prelu1[0], prelu=m.prelu1[1]),
                            dict(x=x12[1], pre=m.w2, bn=m.prelu2[0], prelu=m.prelu2[1])])
        h = self._na(self._lin(_pointwise, ops.cat_channels(ab), c[0]), bn=c[1], prelu=c[2])
        h_pool, h = ops.fanout(h, 2)                                     # squeeze | excite
        gate = ops.se_gate(ops.mean_bc(h_pool), c[3].w1, c[3].w2)
        # the next block starts with a BatchNorm of this output: let the kernel emit its channel sums (train mode)
        return self._na(h, pre=gate, add=bres, add_post=True, emit_stats=tr)

    # ---- FPN.forward, CISTGCN.py:74-79 -------------------------------------------------------------
    def _fpn(self, m, x, x_pool=None):
        blocks = (m.block1, m.block2, m.block3)
        ys = ops.dilated_convs(x, [b[0] for b in blocks])
        outs = self._na_many([dict(x=y, bn=b[1], prelu=b[3]) for y, b in zip(ys, blocks)])   # FPN dropout p = 0 (:533)
        outs.append(ops.mean_bc(x if x_pool is None else x_pool))                  # action context, broadcast below
        return _pointwise(ops.cat_channels(outs, bcast=(False, False, False, True)), m.compress)[0]

    # ---- ContextLayer.forward, CISTGCN.py:463-475 --------------------------------------------------
    def _context(self, m, x7):
        B, To, V, _ = x7.shape
        x = x7.view(B, 1, To, V * 3)
        c1, c2, c3 = m.context_conv1, m.context_conv2, m.context_conv3
        tr = self.training
        w13 = [c[0].weight.view(c[0].out_channels, c[0].in_channels) for c in (c1, c3)]
        if self.fused_context and c1[0].bias is None and c3[0].bias is None and ops.context_heads_ok(x, c1[0].out_channels):
            # heads 1 and 3 (max / mean over the positions of a 1 -> hidden_dim map, BatchNorm, PReLU) from the one-channel sequence
            # itself: their (B, hidden_dim, To, 3V) activations are never stored (csrc/context_heads.hip)
            xa, xb = ops.fanout(x, 2)
            taps = [] if self.act_trace is not None else None
            y1, ym = ops.context_heads(xa, c1, c3, tr, taps=taps)
            if taps is not None:
                self.act_trace[c1[2]], self.act_trace[c3[2]] = (taps[0], None), (taps[1], None)
            y2 = ops.max_bc(self._na(_run_items([_rows_item(xb, c2[0], tr)])[0], bn=c2[1], prelu=c2[2]))
            self._site += 2          # the composite path numbers two more (dropout-free) sites: later layers draw the same masks on either path
            return self._context_tail(m, x7, y1, y2, ym)
        if self.fused_maps and B * To * V * 3 * sum(w.shape[0] for w in w13) >= self.stack_min_elements and ops.pointwise_maps_ok(x, w13):
            # the two 1 -> hidden_dim maps read the sequence once and write their (B, hidden_dim, To, 3V) results from one kernel
            xa, xb = ops.fanout(x, 2)
            y13 = ops.pointwise_maps(xa, w13, tr, biases=[c1[0].bias, c3[0].bias])
            o = [y13[0], _run_items([_rows_item(xb, c2[0], tr)])[0], y13[1]]
        else:
            o = _run_items([_pw_item(x, c1[0], tr), _rows_item(x, c2[0], tr), _pw_item(x, c3[0], tr)])
        r = self._na_many([dict(x=o[i], bn=c[1], prelu=c[2]) for i, c in enumerate((c1, c2, c3))])
        return self._context_tail(m, x7, ops.max_bc(r[0]), ops.max_bc(r[1]), ops.mean_bc(r[2]))

    def _context_tail(self, m, x7, y1, y2, ym):
        """ContextLayer.forward behind the three pooled heads, CISTGCN.py:468-475"""
        B, To, V, _ = x7.shape
        tr = self.training
        o = _run_items([_lin_item(y, h[0], False) for y, h in ((y1, m.map1), (y2, m.map2), (ym, m.map3))])
        heads = self._na_many([dict(x=o[i][0], drop=True, prelu=h[2]) for i, h in enumerate((m.map1, m.map2, m.map3))])
        y = ops.cat_channels(heads)
        o = _run_items([_lin_item(y, m.fmap_s[0], tr), _lin_item(y, m.fmap_t[0], tr)])
        m.joints, m.displacements = self._na_many([dict(x=o[0], bn=m.fmap_s[1], drop=True), dict(x=o[1], bn=m.fmap_t[1], drop=True)])
        m.seq_joints = ops.contract("bt,bv->btv", m.displacements, m.joints)
        n = m.norm_map
        h = self._na(self._lin(_pointwise, m.seq_joints, n[0]), bn=n[1], drop=True, prelu=n[3])
        h_pool, h = ops.fanout(h, 2)
        h = self._na(h, pre=ops.se_gate(ops.mean_bc(h_pool), n[4].w1, n[4].w2))
        h = self._na(self._lin(_pointwise, h, n[5]), bn=n[6], drop=True, prelu=n[8])
        m.seq_joints_n = h
        f = m.fconv
        h = self._na(self._lin(_pointwise, h.view(B, 1, To, V), f[0]), bn=f[1], prelu=f[2])
        h = self._na(self._lin(_pointwise, h, f[3]), bn=f[4], prelu=f[5])
        m.seq_joints_dims = h
        hp_pool, hp = ops.fanout(h.permute(0, 2, 3, 1), 2)
        return self._na(hp, pre=ops.se_gate(ops.mean_bc(hp_pool), m.SE.w1, m.SE.w2))

    # ---- CISTGCN.forward, CISTGCN.py:567-597 -------------------------------------------------------
    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != self.n_input or x.shape[2] != self.n_joints or x.shape[3] != 3:
            raise ValueError("expected input of shape (B, %d, %d, 3), got %s" % (self.n_input, self.n_joints, tuple(x.shape)))
        if torch.jit.is_tracing():
            # `SummaryWriter.add_graph(model, batch)` (train.py:137) traces the model.  The kernels are reached through ctypes and
            # exchange statistics buffers the tracer cannot follow, so the hot path is shown to it as ONE opaque node.
            net = self

            class HotPath(torch.autograd.Function):
                @staticmethod
                def forward(ctx, inp):
                    state = torch._C._get_tracing_state()
                    torch._C._set_tracing_state(None)          # nothing inside is recorded
                    try:
                        with torch.no_grad():
                            return net._forward(inp.detach(), bump_seed=False)[0]
                    finally:
                        torch._C._set_tracing_state(state)

                @staticmethod
                def backward(ctx, g):
                    raise RuntimeError("the traced CISTGCN graph is for display only")

            return HotPath.apply(x),
        return self._forward(x, bump_seed=self.training and self.dropout > 0.0)

    def _forward(self, x, bump_seed):
        ops.begin_step(x.device, bump_seed=bump_seed)
        self._site, self._next_stream = 0, 0
        h = ops.feature_lift(x)                                         # (B,10,T,V)
        block = self._block_staged if self.staged else self._block
        for i, blk in enumerate(self.st_gcnns):
            h = block(blk, h)
            if self.backward_cut is not None and self.backward_cut[0] == i:
                h = self.backward_cut[1](h)                               # runtime.DataParallelStep: two-phase backward
        h = h[0] if isinstance(h, tuple) else h                          # (tensor, channel sums) from a staged block
        h = h.permute(0, 2, 1, 3)                                       # NCTV -> NTCV (view)
        ha = ops.fanout(h, 2)                                           # dilated convolutions | pooled context
        z = self._na(self._fpn(self.txcnns[0], ha[0], ha[1]), prelu=self.prelus[0])
        for i in range(1, self.n_txcnn_layers):
            za = ops.fanout(z, 3)                                       # ... | skip connection
            z = self._na(self._fpn(self.txcnns[i], za[0], za[1]), prelu=self.prelus[i], add=za[2], add_post=True)
        d = self.dim_conversor
        z = self._na(self._lin(_pointwise, z.permute(0, 2, 1, 3), d[0]), bn=d[1], prelu=d[2])
        z = self._na(_pointwise(z, d[3])[0], prelu=d[4])                # PReLU(3): per-channel slopes (:545)
        x7, x7o = ops.fanout(ops.cumsum_time(z.permute(0, 2, 3, 1)), 2)   # (B,T_out,V,3): context branch | output block
        def output_blocks():
            x8 = x7o.permute(0, 3, 2, 1)                                # (B,3,V,T_out) view
            for blk in self.st_gcnns_o:
                x8 = block(blk, x8)
            return x8[0] if isinstance(x8, tuple) else x8
        act, x8 = self._parallel([lambda: self._context(self.context_layer, x7), output_blocks], [x7])
        return ops.add3(x[:, -1:], x8.permute(0, 3, 2, 1), act),
