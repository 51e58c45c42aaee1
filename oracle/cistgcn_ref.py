"""CPU oracle for the CIST-GCN forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  This file is a from-scratch stock-PyTorch restatement of the
reference algorithm (QualityMinds/cistgcn, `human_motion_prediction/models/CISTGCN/CISTGCN.py`
and `models/layers/SE.py`).  It is the checker for the HIP path and the `cpu_baseline` of
`bench.py`; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import it.  The product package `cistgcn_amd` never imports anything from `oracle/`.

Parity status: PINNED.  `tools/gen_golden.py` imports the real reference in the build
container and writes input/output/gradient vectors to `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this file against every one of them (eval mode and
train mode with dropout 0, four (T_in, V) shapes).

The parameter tree (names, shapes, registration order, init) equals the reference's so that
a reference `state_dict` loads unchanged.  The arithmetic is written out
stage by stage; every function cites the reference lines it follows.
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# parameter tree helpers
# ----------------------------------------------------------------------------------------------
class Slots(nn.Module):
    """Children registered under explicit integer slots -> state_dict keys '<slot>.<name>'.

    The reference uses nn.Sequential; parameter-free stages (dropout, ReLU, sigmoid) occupy a
    slot number but contribute no key, so the slots listed here have gaps.
    """

    def __init__(self, slots):
        super().__init__()
        for idx, mod in slots:
            self.add_module(str(idx), mod)

    def __getitem__(self, idx):
        return self._modules[str(idx)]


def _xavier_small(mod, gain, convs=False):
    # CISTGCN.py:175-181 (Map2Adj: gain .05, convs too) and :559-565 (CISTGCN: gain .1, conv branch commented out)
    for m in mod.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight, gain=gain)
        if convs and isinstance(m, (nn.Conv2d, nn.Conv1d)):
            nn.init.xavier_normal_(m.weight, gain=gain)
        if isinstance(m, nn.PReLU):
            nn.init.constant_(m.weight, 0.25)


def _tower(cin, mid, kernel, cout):
    # CISTGCN.py:138-153
    return Slots([(0, nn.Conv2d(cin, mid, 1, bias=False)), (1, nn.BatchNorm2d(mid)), (2, nn.PReLU()),
                  (3, nn.Conv2d(mid, mid, kernel, bias=False)), (4, nn.BatchNorm2d(mid)),
                  (6, nn.Conv2d(mid, cout, 1, bias=False))])


class Map2AdjParams(nn.Module):
    def __init__(self, cin, T, V, domain):
        super().__init__()
        mid = cin // 2
        self.time_compress = _tower(cin, mid, (T, 1), T)
        self.joint_compress = _tower(cin, mid, (1, V), V)
        ch = V if domain == "space" else T
        self.expansor = Slots([(0, nn.Conv2d(ch, ch, 1, bias=False)), (1, nn.BatchNorm2d(ch)),
                               (3, nn.PReLU()), (4, nn.Conv2d(ch, ch, 1, bias=False))])
        for part in (self.time_compress, self.joint_compress, self.expansor):
            _xavier_small(part, 0.05, convs=True)


class GraphParam(nn.Module):
    """Non-interpretable adjacency (CISTGCN.py:104-120); never selected by the shipped YAMLs."""

    def __init__(self, T, V, domain, interpretable):
        super().__init__()
        if not interpretable:
            shape = (T, V, V) if domain == "time" else (V, T, T)
            self.A = nn.Parameter(torch.empty(shape))
            bound = 1.0 / (shape[1] ** 0.5)
            self.A.data.uniform_(-bound, bound)


class DomainParams(nn.Module):
    # CISTGCN.py:208-257 (registration order: gcn, tcn, residual, map_to_adj, prelu)
    def __init__(self, cin, cout, T, V, domain, interpretable):
        super().__init__()
        self.domain, self.interpretable = domain, interpretable
        self.gcn = GraphParam(T, V, domain, interpretable)
        self.tcn = Slots([(0, nn.Conv2d(cin, cout, 1)), (1, nn.BatchNorm2d(cout))])
        if cin != cout:
            self.residual = Slots([(0, nn.Conv2d(cin, cout, 1)), (1, nn.BatchNorm2d(cout))])
        else:
            self.residual = nn.Identity()
        self.map_to_adj = Map2AdjParams(cin, T, V, domain) if interpretable else nn.Identity()
        self.prelu = nn.PReLU()


class SEParams(nn.Module):
    # SE.py:5-20 / 24-41 (1d: c//r ; 2d: max(1, c//r))
    def __init__(self, c, reduction, two_d):
        super().__init__()
        hid = c // reduction
        if two_d and hid < 1:
            hid = 1
        self.excitation = Slots([(0, nn.Linear(c, hid, bias=False)), (2, nn.Linear(hid, c, bias=False))])


def _gate_conv(cin, mid, cout, T, V):
    # CISTGCN.py:323-340
    return Slots([(0, nn.Conv2d(cin, mid, (T, 1), bias=False)), (1, nn.BatchNorm2d(mid)), (3, nn.PReLU()),
                  (4, nn.Conv2d(mid, cout, (1, V), bias=False)), (5, nn.BatchNorm2d(cout)), (7, nn.PReLU())])


def _gate_map(cin, cout):
    # CISTGCN.py:341-352
    return Slots([(0, nn.Linear(cin, cout, bias=False)), (1, nn.BatchNorm1d(cout)), (3, nn.PReLU()),
                  (4, nn.Linear(cout, cout, bias=False))])


class BlockParams(nn.Module):
    # CISTGCN.py:289-358
    def __init__(self, cin, cout, interpretable, T, V, reduction):
        super().__init__()
        self.dsgn = DomainParams(cin, cout, T, V, "space", interpretable)
        self.tsgn = DomainParams(cin, cout, T, V, "time", interpretable)
        self.compressor = Slots([(0, nn.Conv2d(2 * cout, cout, 1, bias=False)), (1, nn.BatchNorm2d(cout)),
                                 (2, nn.PReLU()), (3, SEParams(cout, reduction, True))])
        if cin != cout:
            self.residual = Slots([(0, nn.Conv2d(cin, cout, 1)), (1, nn.BatchNorm2d(cout))])
        else:
            self.residual = nn.Identity()
        mid = cout // 2 if cout // 2 > 1 else 1
        self.global_norm = nn.BatchNorm2d(cin)
        self.conv_s = _gate_conv(cin, mid, cout, T, V)
        self.conv_t = _gate_conv(cin, mid, cout, T, V)
        self.map_s = _gate_map(cout + 2 + 2 * T, cout)
        self.map_t = _gate_map(cout + 2 + 2 * T, cout)
        self.prelu1 = Slots([(0, nn.BatchNorm2d(cout)), (1, nn.PReLU())])
        self.prelu2 = Slots([(0, nn.BatchNorm2d(cout)), (1, nn.PReLU())])


class FPNParams(nn.Module):
    # CISTGCN.py:38-72 ; square kernel k, dilations 1,2,3 with "same" padding
    def __init__(self, cin, cout, k):
        super().__init__()
        p = (k - 1) // 2
        self.dil = (1, 1 + p, 1 + 2 * p)
        self.pad = (p, 2 * p, 3 * p)
        for i in range(3):
            self.add_module("block%d" % (i + 1), Slots([
                (0, nn.Conv2d(cin, cout, k, padding=self.pad[i], dilation=self.dil[i])),
                (1, nn.BatchNorm2d(cout)), (3, nn.PReLU())]))
        self.compress = nn.Conv2d(3 * cout + cin, cout, 1)


class ContextParams(nn.Module):
    # CISTGCN.py:394-461
    def __init__(self, hidden, To, V, reduction):
        super().__init__()
        def cc(kernel):
            return Slots([(0, nn.Conv2d(1, hidden, kernel, bias=False)), (1, nn.BatchNorm2d(hidden)), (2, nn.PReLU())])
        self.context_conv1, self.context_conv2, self.context_conv3 = cc(1), cc((To, 1)), cc(1)
        def mp():
            return Slots([(0, nn.Linear(hidden, To, bias=False)), (2, nn.PReLU())])
        self.map1, self.map2, self.map3 = mp(), mp(), mp()
        self.fmap_s = Slots([(0, nn.Linear(3 * To, V, bias=False)), (1, nn.BatchNorm1d(V))])
        self.fmap_t = Slots([(0, nn.Linear(3 * To, To, bias=False)), (1, nn.BatchNorm1d(To))])
        self.norm_map = Slots([(0, nn.Conv1d(To, To, 1, bias=False)), (1, nn.BatchNorm1d(To)), (3, nn.PReLU()),
                               (4, SEParams(To, reduction, False)),
                               (5, nn.Conv1d(To, To, 1, bias=False)), (6, nn.BatchNorm1d(To)), (8, nn.PReLU())])
        self.fconv = Slots([(0, nn.Conv2d(1, 3, 1, bias=False)), (1, nn.BatchNorm2d(3)), (2, nn.PReLU()),
                            (3, nn.Conv2d(3, 3, 1, bias=False)), (4, nn.BatchNorm2d(3)), (5, nn.PReLU())])
        self.SE = SEParams(To, reduction, True)


# ----------------------------------------------------------------------------------------------
# arithmetic helpers
# ----------------------------------------------------------------------------------------------
def _bn(x, m, train):
    if train and m.num_batches_tracked is not None:
        m.num_batches_tracked += 1
    return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, train, m.momentum, m.eps)


def _conv(x, m):
    if isinstance(m, nn.Conv1d):
        return F.conv1d(x, m.weight, m.bias)
    return F.conv2d(x, m.weight, m.bias, padding=m.padding, dilation=m.dilation)


def _act(x, m):
    return F.prelu(x, m.weight)


def _se(x, m):
    # SE.py:16-20, 37-41 : mean over everything behind the channel axis -> 2-layer gate -> scale
    pooled = x.flatten(2).mean(-1)
    gate = torch.sigmoid(F.linear(F.relu(F.linear(pooled, m.excitation[0].weight)), m.excitation[2].weight))
    return x * gate.reshape(gate.shape + (1,) * (x.dim() - 2))


class CISTGCN(nn.Module):
    """Oracle model; ctor signature, class name, state_dict keys and post-forward attributes
    follow CISTGCN.py:478-597."""

    def __init__(self, arch, learn):
        super().__init__()
        p = arch.model_params
        self.clipping = p.clipping
        self.n_input, self.n_output, self.n_joints = p.input_n, p.output_n, p.joints
        self.n_txcnn_layers = p.n_txcnn_layers
        self.reduction, self.hidden_dim = p.reduction, p.hidden_dim
        self.dropout = float(learn.dropout)
        # the reference mutates the config lists in place (:514-517,:548); work on copies
        widths = [10] + list(copy.copy(p.input_gcn.model_complexity)) + [10]
        interp = list(p.input_gcn.interpretable)
        widths_o = [3] + list(copy.copy(p.output_gcn.model_complexity))
        interp_o = list(p.output_gcn.interpretable)
        T, V, To = self.n_input, self.n_joints, self.n_output

        self.st_gcnns = nn.ModuleList()
        self.txcnns = nn.ModuleList()
        self.se = nn.ModuleList()
        self.in_conv = nn.ModuleList()
        self.context_layer = nn.ModuleList()
        self.trans = nn.ModuleList()
        for i in range(len(widths) - 1):
            self.st_gcnns.append(BlockParams(widths[i], widths[i + 1], interp[i], T, V, self.reduction))
        self.context_layer = ContextParams(self.hidden_dim, To, V, self.reduction)
        self.txcnns.append(FPNParams(T, To, p.txc_kernel_size))
        for _ in range(1, self.n_txcnn_layers):
            self.txcnns.append(FPNParams(To, To, p.txc_kernel_size))
        self.prelus = nn.ModuleList([nn.PReLU() for _ in range(self.n_txcnn_layers)])
        self.dim_conversor = Slots([(0, nn.Conv2d(10, 3, 1, bias=False)), (1, nn.BatchNorm2d(3)), (2, nn.PReLU()),
                                    (3, nn.Conv2d(3, 3, 1, bias=False)), (4, nn.PReLU(3))])
        self.st_gcnns_o = nn.ModuleList()
        for i in range(len(widths_o) - 1):
            # NB: time_dim=V, joints_dim=T_out for the output block (:553)
            self.st_gcnns_o.append(BlockParams(widths_o[i], widths_o[i + 1], interp_o[i], V, To, self.reduction))
        for part in (self.st_gcnns_o, self.st_gcnns, self.txcnns):
            _xavier_small(part, 0.1)

    # ---- row A: CISTGCN.py:568-577 -----------------------------------------------------------
    @staticmethod
    def feature_lift(x):
        vel = torch.cat((x[:, 1:] - x[:, :-1], x[:, -1:]), dim=1)
        acc = torch.cat((vel[:, 1:] - vel[:, :-1], vel[:, -1:]), dim=1)
        speed = torch.norm(vel, dim=-1, keepdim=True)
        return torch.cat((x, acc, vel, speed), dim=-1).permute(0, 3, 1, 2)

    # Dropout sites (the reference's nn.Dropout(p, inplace=True) layers, CISTGCN.py:143, 150, 167, 236, 327, 333, 346, 424-436, 440-449).
    # `site` names the site: the BatchNorm in front of it, or - where there is none (ContextLayer.map1-3) - the PReLU behind it.
    # `drop_hook` (tests only): callable (x, site) -> dropped x, lets a checker apply the very masks another implementation drew.
    drop_hook = None

    def _drop(self, x, site=None):
        if self.drop_hook is not None and self.training and self.dropout > 0.0:
            return self.drop_hook(x, site)
        return F.dropout(x, self.dropout, self.training)

    # ---- row G: CISTGCN.py:183-189 -----------------------------------------------------------
    def map2adj(self, m, x):
        tr = self.training
        def tower(t):
            h = _act(_bn(_conv(x, t[0]), t[1], tr), t[2])
            h = self._drop(_bn(_conv(h, t[3]), t[4], tr), t[4])
            return _conv(h, t[6])
        q = tower(m.time_compress)    # (B, T, 1, V)
        s = tower(m.joint_compress)   # (B, V, T, 1)
        return q, s

    def adjacency(self, layer, x):
        m = layer.map_to_adj
        q, s = self.map2adj(m, x)
        if layer.domain == "space":
            o = s * q.permute(0, 3, 2, 1)              # o[b,v,t,tau] = s[b,v,t] q[b,tau,v]
        else:
            o = s.permute(0, 2, 1, 3) * q              # o[b,t,v,w]  = s[b,v,t] q[b,t,w]
        e = m.expansor
        h = _act(self._drop(_bn(_conv(o, e[0]), e[1], self.training), e[1]), e[3])
        return _conv(h, e[4])

    # ---- rows E,F: CISTGCN.py:259-269, 122-124 -------------------------------------------------
    def domain_layer(self, layer, x):
        tr = self.training
        res = x if isinstance(layer.residual, nn.Identity) else _bn(_conv(x, layer.residual[0]), layer.residual[1], tr)
        if layer.interpretable:
            A = self.adjacency(layer, x)
            layer.Adj = A
            if layer.domain == "space":   # per (n,v): (C x T)(T x T)
                g = torch.matmul(x.permute(0, 3, 1, 2), A).permute(0, 2, 3, 1)
            else:                          # per (n,t): (C x V)(V x V)
                g = torch.matmul(x.permute(0, 2, 1, 3), A).permute(0, 2, 1, 3)
        else:
            layer.Adj = x
            eq = "nctv,vtq->ncqv" if layer.domain == "space" else "nctv,tvw->nctw"
            g = torch.einsum(eq, x, layer.gcn.A)
        y = self._drop(_bn(_conv(g.contiguous(), layer.tcn[0]), layer.tcn[1], tr), layer.tcn[1])
        return _act(y + res, layer.prelu)

    # ---- row C: CISTGCN.py:360-371 -----------------------------------------------------------
    @staticmethod
    def block_stats(xn):
        return torch.cat((xn.mean((3, 2)).mean(1, keepdim=True), xn.mean(3).mean(1),
                          xn.std((3, 2)).std(1, keepdim=True), xn.std(3).std(1)), dim=1)

    # ---- row D: CISTGCN.py:378-384 -----------------------------------------------------------
    def gate(self, conv, mp, xn, stats):
        tr = self.training
        h = _act(self._drop(_bn(_conv(xn, conv[0]), conv[1], tr), conv[1]), conv[3])
        h = _act(self._drop(_bn(_conv(h, conv[4]), conv[5], tr), conv[5]), conv[7])
        h = torch.cat((h.flatten(1), stats), dim=1)
        h = _act(self._drop(_bn(F.linear(h, mp[0].weight), mp[1], tr), mp[1]), mp[3])
        return F.linear(h, mp[4].weight)

    # ---- row B: CISTGCN.py:373-390 -----------------------------------------------------------
    def block(self, m, x):
        tr = self.training
        xn = _bn(x, m.global_norm, tr)
        stats = self.block_stats(xn)
        m.w1 = self.gate(m.conv_s, m.map_s, xn, stats)
        m.w2 = self.gate(m.conv_t, m.map_t, xn, stats)
        x1 = self.domain_layer(m.dsgn, xn)
        x2 = self.domain_layer(m.tsgn, xn)
        a = _act(_bn(m.w1[:, :, None, None] * x1, m.prelu1[0], tr), m.prelu1[1])
        b = _act(_bn(m.w2[:, :, None, None] * x2, m.prelu2[0], tr), m.prelu2[1])
        c = m.compressor
        h = _se(_act(_bn(_conv(torch.cat((a, b), dim=1), c[0]), c[1], tr), c[2]), c[3])
        res = xn if isinstance(m.residual, nn.Identity) else _bn(_conv(xn, m.residual[0]), m.residual[1], tr)
        return h + res

    # ---- row I: CISTGCN.py:74-79 -------------------------------------------------------------
    def fpn(self, m, x):
        tr = self.training
        outs = []
        for name in ("block1", "block2", "block3"):
            b = getattr(m, name)
            outs.append(_act(_bn(_conv(x, b[0]), b[1], tr), b[3]))   # dropout p=0 (:533)
        outs.append(x.mean((2, 3), keepdim=True).expand(-1, -1, x.shape[2], x.shape[3]))
        return _conv(torch.cat(outs, dim=1), m.compress)

    # ---- row K: CISTGCN.py:463-475 -----------------------------------------------------------
    def context(self, m, x):
        tr = self.training
        b = x.shape[0]
        To, V = self.n_output, self.n_joints
        def cc(c):
            return _act(_bn(_conv(x, c[0]), c[1], tr), c[2])
        y1 = cc(m.context_conv1).flatten(2).max(-1)[0]
        y2 = cc(m.context_conv2).flatten(2).max(-1)[0]
        ym = cc(m.context_conv3).mean((2, 3))
        def mp(y, mm):
            return _act(self._drop(F.linear(y, mm[0].weight), mm[2]), mm[2])
        y = torch.cat((mp(y1, m.map1), mp(y2, m.map2), mp(ym, m.map3)), dim=1)
        m.joints = self._drop(_bn(F.linear(y, m.fmap_s[0].weight), m.fmap_s[1], tr), m.fmap_s[1])
        m.displacements = self._drop(_bn(F.linear(y, m.fmap_t[0].weight), m.fmap_t[1], tr), m.fmap_t[1])
        m.seq_joints = m.displacements[:, :, None] * m.joints[:, None, :]
        n = m.norm_map
        h = _act(self._drop(_bn(_conv(m.seq_joints, n[0]), n[1], tr), n[1]), n[3])
        h = _se(h, n[4])
        h = _act(self._drop(_bn(_conv(h, n[5]), n[6], tr), n[6]), n[8])
        m.seq_joints_n = h
        f = m.fconv
        h = _act(_bn(_conv(h.view(b, 1, To, V), f[0]), f[1], tr), f[2])
        h = _act(_bn(_conv(h, f[3]), f[4], tr), f[5])
        m.seq_joints_dims = h
        return _se(h.permute(0, 2, 3, 1), m.SE)

    # ---- CISTGCN.py:567-597 ------------------------------------------------------------------
    def forward(self, x):
        b, _, V, _ = x.shape
        tr = self.training
        h = self.feature_lift(x)
        for blk in self.st_gcnns:
            h = self.block(blk, h)
        h = h.permute(0, 2, 1, 3)                                 # NCTV -> NTCV
        z = _act(self.fpn(self.txcnns[0], h), self.prelus[0])
        for i in range(1, self.n_txcnn_layers):
            z = _act(self.fpn(self.txcnns[i], z), self.prelus[i]) + z
        d = self.dim_conversor                                   # row J
        z = z.permute(0, 2, 1, 3)
        z = _act(_bn(_conv(z, d[0]), d[1], tr), d[2])
        z = _act(_conv(z, d[3]), d[4]).permute(0, 2, 3, 1)       # (B, To, V, 3)
        x7 = z.cumsum(1)
        act = self.context(self.context_layer, x7.reshape(b, 1, self.n_output, V * 3))
        x8 = x7.permute(0, 3, 2, 1)                               # (B, 3, V, To)
        for blk in self.st_gcnns_o:
            x8 = self.block(blk, x8)
        return x[:, -1:] + (x8.permute(0, 3, 2, 1) + act),


def mpjpe(pred, target):
    """Training loss actually used: losses/losses.py:50-61 with reduce_axis=[] (full mean)."""
    return torch.norm(pred - target, 2, dim=-1).mean()
