"""Host-side operators of the CIST-GCN path: thin `torch.autograd.Function`s over the C ABI of
libcistgcn_hip.so.  PyTorch supplies device memory, streams and the autograd tape; every number
is produced by a HIP kernel.  Views (permute / reshape / slicing) are metadata only: all kernels
take strides, so layout changes of the reference (`CISTGCN.py:582,588,592,595`) cost nothing.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import NormAct, View4

_vp = ctypes.c_void_p


# ----------------------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------------------
def _ptr(t):
    return _vp(t.data_ptr()) if t is not None else None


def _stream(t):
    if t.is_cuda:
        return _vp(torch.cuda.current_stream(t.device).cuda_stream)
    if not _lib._host_pointers_ok:
        raise RuntimeError("cistgcn_amd operators run on MI355X only: got a %s tensor (no CPU path exists)" % t.device)
    return None


def _chk(t, name="tensor"):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t


def _view4(t):
    """(sizes, strides) of a 2/3/4-D tensor as a CgView4 (missing trailing dims have size 1)."""
    if t.dim() < 2 or t.dim() > 4:
        raise ValueError("expected a 2..4-D tensor, got %d-D" % t.dim())
    n = list(t.shape) + [1] * (4 - t.dim())
    s = list(t.stride()) + [0] * (4 - t.dim())
    v = View4()
    for i in range(4):
        v.n[i] = n[i]
        v.s[i] = s[i]
    return v


class _Arena:
    """Per-device f64 scratch that is all-zero at the start of every step: BatchNorm sums and
    backward reductions take slices from it; one memset per step replaces one per layer."""

    def __init__(self, device, n=1 << 21):
        self.buf = torch.zeros(n, dtype=torch.float64, device=device)
        self.cur = 0
        self.high = 0

    def begin_step(self):
        if self.high:
            _lib.call("cg_zero", _ptr(self.buf), self.high * 8, _stream(self.buf))
        self.cur = 0

    def take(self, n):
        if self.cur + n > self.buf.numel():
            raise RuntimeError("cistgcn_amd: step scratch exhausted (missing ops.begin_step()?)")
        s = self.buf[self.cur:self.cur + n]
        if _ARENA_CHECK and float(s.abs().sum()) != 0.0:      # debugging aid: a slice must be all-zero when handed out
            raise RuntimeError("cistgcn_amd: dirty scratch slice at %d (+%d): %s" % (self.cur, n, s.tolist()))
        self.cur += (n + 1) & ~1
        self.high = max(self.high, self.cur)
        return s


class _ZeroPool:
    """Per-device f32 scratch that is all-zero at the start of every step (opt-in, see `step_scratch`): split-K /
    accumulated contraction outputs, zero halos and replicated dW accumulators are carved from it, so that one
    memset per step replaces one per launch (~110 graph nodes of the B=16 step).  A slice is valid until the next
    `begin_step`; tensors that must outlive the step (parameter gradients) have to be consumed before it."""

    def __init__(self, device, n):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.cur = 0
        self.high = 0

    def begin_step(self):
        if self.high:
            _lib.call("cg_zero", _ptr(self.buf), self.high * 4, _stream(self.buf))
        self.cur = 0

    def take(self, n):
        n64 = (n + 63) & ~63                 # 256-byte slots
        if self.cur + n64 > self.buf.numel():
            return None
        s = self.buf[self.cur:self.cur + n]
        self.cur += n64
        self.high = max(self.high, self.cur)
        return s


_arenas = {}
_zero_pools = {}
_seeds = {}
_ARENA_CHECK = bool(int(__import__("os").environ.get("CISTGCN_ARENA_CHECK", "0")))


def _dev(device):
    """Canonical device key ("cuda" and "cuda:0" must name the same scratch and seed)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def _arena(device):
    device = _dev(device)
    a = _arenas.get(device)
    if a is None:
        a = _arenas[device] = _Arena(device)
    return a


def step_scratch(device, enable=True, floats=1 << 25):
    """Opt in (or out) of a per-step zero pool on `device` for the whole process.  Only for callers that run exactly one
    forward + backward per `begin_step` and consume the gradients before the next one: gradients of that step may alias
    pool memory, which the next `begin_step` clears.  The step runtime does NOT use this switch: it owns a private pool and
    installs it only around its own step (`step_pool`), so no other forward in the process ever sees pool-backed gradients."""
    device = _dev(device)
    if enable:
        if device not in _zero_pools:
            _zero_pools[device] = _ZeroPool(device, int(floats))
    else:
        _zero_pools.pop(device, None)


def new_step_pool(device, floats=1 << 25):
    """A private zero pool (see `step_pool`)."""
    return _ZeroPool(_dev(device), int(floats))


class step_pool:
    """Context manager: `pool` is the zero pool of its device inside the block and only there.  runtime.GraphedStep /
    EagerStep wrap each of their own forward + backward passes in it; parameter gradients produced inside may alias the
    pool and are valid until that object's next step (it gathers or consumes them right after the pass)."""

    def __init__(self, pool):
        self.pool = pool

    def __enter__(self):
        dev = self.pool.buf.device
        self.prev = _zero_pools.get(dev)
        _zero_pools[dev] = self.pool
        return self.pool

    def __exit__(self, *exc):
        dev = self.pool.buf.device
        if self.prev is None:
            _zero_pools.pop(dev, None)
        else:
            _zero_pools[dev] = self.prev
        return False


def _zeros(n, device):
    """(flat zero-filled f32 tensor of n elements, True if it came from the step pool)"""
    pool = _zero_pools.get(_dev(device)) if _zero_pools else None
    if pool is not None:
        s = pool.take(n)
        if s is not None:
            return s, True
    z = torch.empty(n, dtype=torch.float32, device=device)
    _lib.call("cg_zero", _ptr(z), n * 4, _stream(z))
    return z, False


def seed_state(device):
    device = _dev(device)
    s = _seeds.get(device)
    if s is None:
        s = _seeds[device] = torch.full((1,), 0x5DEECE66D, dtype=torch.int64, device=device)
    return s


# Dropout keep-masks are regenerated in backward from the device-resident seed word, so a backward pass must see the seed of
# its own forward.  Every change of the word is counted per device; a forward with dropout records the count and its backward
# raises if the seed has moved in between (two micro-batches before one backward, a retained graph, ...) instead of silently
# applying different masks.  Under HIP-graph replay nothing of this runs: a captured step contains its own bump + fwd + bwd.
_seed_epochs = {}


def seed_epoch(device):
    return _seed_epochs.get(_dev(device), 0)


def _check_seed_epoch(ctx, device, what):
    ep = getattr(ctx, "seed_epoch", None)
    if ep is not None and ep != seed_epoch(device):
        raise RuntimeError("%s: the dropout seed advanced between this forward and its backward (another training forward ran in "
                           "between); run backward before the next forward or use drop_p = 0" % what)


def manual_seed(seed, device):
    seed_state(device).fill_(int(seed))
    _seed_epochs[_dev(device)] = seed_epoch(device) + 1


def begin_step(device, bump_seed=False):
    """Start of a forward pass: re-zero the statistics scratch, optionally advance the dropout seed."""
    device = _dev(device)
    _arena(device).begin_step()
    if _zero_pools:
        pool = _zero_pools.get(device)
        if pool is not None:
            pool.begin_step()
    if bump_seed:
        s = seed_state(device)
        _lib.call("cg_seed_bump", _ptr(s), _stream(s))
        _seed_epochs[device] = _seed_epochs.get(device, 0) + 1


# ----------------------------------------------------------------------------------------------
# generic contraction
# ----------------------------------------------------------------------------------------------
class _Plan:
    __slots__ = ("tables", "G", "M", "N", "K", "splitk", "a_kfast", "x_kfast", "dense", "x_vec", "mode", "ws_floats")


_plans = {}


def _offsets(labels, sizes, strides):
    off = np.zeros(1, dtype=np.int64)
    for l in labels:
        off = (off[:, None] + (np.arange(sizes[l], dtype=np.int64) * strides.get(l, 0))[None, :]).reshape(-1)
    return off


def _min_stride(labels, sizes, strides):
    vals = [abs(strides.get(l, 0)) for l in labels if sizes[l] > 1]
    return min(vals) if vals else None


def _plan(sizes, la, lx, ly, sa, sx, sy, oa, ox, oy, bias_label, device, y_dense, aligned=False):
    key = (tuple(sorted(sizes.items())), la, lx, ly, tuple(sorted(sa.items())), tuple(sorted(sx.items())),
           tuple(sorted(sy.items())), oa, ox, oy, bias_label, str(device), y_dense, aligned)
    p = _plans.get(key)
    if p is not None:
        return p
    A, X, Y = set(la), set(lx), set(ly)
    if (A | X) - (A & X) - Y or Y - (A | X):
        raise ValueError("contract: every index must appear in two of the three operands (%s,%s->%s)" % (la, lx, ly))
    g = [l for l in ly if l in A and l in X]
    m = [l for l in ly if l in A and l not in X]
    n = [l for l in ly if l in X and l not in A]
    k = [l for l in lx if l in A and l not in Y]
    tabs = [_offsets(g, sizes, sa) + oa, _offsets(g, sizes, sx) + ox, _offsets(g, sizes, sy) + oy,
            _offsets(m, sizes, sa), _offsets(m, sizes, sy),
            _offsets(m, sizes, {bias_label: 1} if bias_label else {}),
            _offsets(n, sizes, sx), _offsets(n, sizes, sy),
            _offsets(k, sizes, sa), _offsets(k, sizes, sx)]
    flat = np.concatenate(tabs)
    if flat.size and (flat.max() >= 2 ** 31 or flat.min() < -2 ** 31):
        raise ValueError("contract: tensor too large for int32 offsets")
    p = _Plan()
    p.G, p.M, p.N, p.K = tabs[0].size, tabs[3].size, tabs[6].size, tabs[8].size
    p.tables = torch.from_numpy(flat.astype(np.int32)).to(device)
    ka, ma = _min_stride(k, sizes, sa), _min_stride(m, sizes, sa)
    kx, nx = _min_stride(k, sizes, sx), _min_stride(n, sizes, sx)
    p.a_kfast = 1 if (ma is None or (ka is not None and ka < ma)) else 0
    p.x_kfast = 1 if (nx is None or (kx is not None and kx < nx)) else 0
    # X readable as float4 along n: n-offsets contiguous in aligned groups of four and every other offset a multiple of 4
    nx_tab = tabs[6]
    p.x_vec = 0
    if not p.x_kfast and p.N % 4 == 0 and p.N >= 4:
        quads = nx_tab.reshape(-1, 4)
        if (np.diff(quads, axis=1) == 1).all() and (quads[:, 0] % 4 == 0).all() and (tabs[1] % 4 == 0).all() and (tabs[9] % 4 == 0).all():
            p.x_vec = 1
    blocks = p.G * ((p.M + (15 if p.M <= 16 else 63)) // (16 if p.M <= 16 else 64)) * ((p.N + 63) // 64)
    p.splitk = 1
    p.dense = y_dense
    p.mode, p.ws_floats = 0, 0
    if y_dense and p.K >= 256 and blocks < 512 and aligned and _KRED and _kred_ok(p, tabs):
        # weight gradient of a pointwise map: K-reduction kernel (float4 along k straight into the matrix cores,
        # replicated partial sums folded by the last workgroup)
        bt = 16 if max(p.M, p.N) <= 16 else 32
        tiles = p.G * ((p.M + bt - 1) // bt) * ((p.N + bt - 1) // bt)
        p.mode = 2
        p.splitk = int(max(1, (p.K + 4063) // 4064, min(512, (p.K + 255) // 256, max(1, _KRED_BLOCKS // tiles))))
        p.ws_floats = int(_lib.lib().cg_contract_kred_ws_floats(p.G, p.M, p.N))
    elif (y_dense and aligned and _STREAM and p.x_vec and p.K <= 128 and p.N >= _STREAM_MIN_N and _quads(tabs[7])
          and (tabs[4] % 4 == 0).all() and (tabs[2] % 4 == 0).all()):
        p.mode = 1        # pointwise map over a long contiguous position axis: streaming kernel (A panel in LDS, X via float4)
    elif y_dense and p.K >= 256 and blocks < 512:
        # few output tiles and a long reduction (weight / bias gradients): spread K over workgroups
        # every split adds its tile with fp32 atomics: splitk blocks serialise on each output address (~0.1 us each on
        # MI355X) while each block walks K/splitk in ~1 us steps of 16 -> balance at splitk ~ sqrt(K/2)
        p.splitk = int(max(1, min((p.K + 63) // 64, (1024 + blocks - 1) // blocks, int((p.K / 2.0) ** 0.5))))
    _plans[key] = p
    return p


_KRED = bool(int(__import__("os").environ.get("CISTGCN_KRED", "1")))     # tuning aid: 0 = always the tiled split-K path
_STREAM = bool(int(__import__("os").environ.get("CISTGCN_STREAM", "1")))  # tuning aid: 0 = tiled path for pointwise maps
# below ~6e4 positions a launch is latency bound either way and splitting a batch into two kernels costs a graph node
_STREAM_MIN_N = int(__import__("os").environ.get("CISTGCN_STREAM_MIN_N", "65536"))
_KRED_BLOCKS = int(__import__("os").environ.get("CISTGCN_KRED_BLOCKS", "128"))
_KRED_MAX = int(__import__("os").environ.get("CISTGCN_KRED_MAX", "64"))
# measured on MI355X (A/B on one box): below ~1e5 reduction elements the tiled split-K plan is as fast or faster
# (B=16 step +2.5 % with K-reduction everywhere), above it the K-reduction kernel wins (B=256, C=64 step -3.3 %)
_KRED_MIN_K = int(__import__("os").environ.get("CISTGCN_KRED_MIN_K", "65536"))


def _quads(tab):
    """offsets contiguous in 16-byte aligned groups of four"""
    if tab.size % 4:
        return False
    q = tab.reshape(-1, 4)
    return bool((np.diff(q, axis=1) == 1).all() and (q[:, 0] % 4 == 0).all())


def _kred_ok(p, tabs):
    if p.K % 4 or p.K < _KRED_MIN_K or p.G * p.M * p.N > 65536 or max(p.M, p.N) > _KRED_MAX:
        return False
    return (_quads(tabs[8]) and _quads(tabs[9]) and all((tabs[i] % 4 == 0).all() for i in (0, 1, 3, 6)))


def _label_strides(t, labels):
    if t.dim() != len(labels):
        raise ValueError("contract: operand has %d dims for labels '%s'" % (t.dim(), labels))
    return {l: s for l, s in zip(labels, t.stride())}


class _Prep:
    """One contraction ready to launch: descriptor, output tensor, optional channel-sum slice."""
    __slots__ = ("desc", "y", "stats", "zero", "tag", "ws_floats", "chained")


def _contract_prepare(spec, a, x, bias=None, bias_label=None, sizes=None, sa=None, sx=None, oa=0, ox=0, stats_label=None,
                      out=None):
    ins, ly = spec.split("->")
    la, lx = ins.split(",")
    _chk(a, "a"), _chk(x, "x")
    if sizes is None:
        sizes = {}
        for t, ls in ((a, la), (x, lx)):
            for l, n in zip(ls, t.shape):
                if sizes.setdefault(l, n) != n:
                    raise ValueError("contract: size mismatch for index '%s' in %s" % (l, spec))
    sa = _label_strides(a, la) if sa is None else sa
    sx = _label_strides(x, lx) if sx is None else sx
    shape = [sizes[l] for l in ly]
    y = out.view(shape) if out is not None else torch.empty(shape, dtype=torch.float32, device=x.device)
    sy = _label_strides(y, ly)
    p = _plan(sizes, la, lx, ly, sa, sx, sy, oa, ox, 0, bias_label or stats_label, x.device, True,
              a.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)
    r = _Prep()
    kred = p.mode == 2
    r.y, r.zero, r.stats, r.tag, r.chained = y, (p.splitk > 1 and not kred), None, spec, False
    r.ws_floats = p.ws_floats
    if stats_label is not None and p.splitk == 1 and p.mode != 2:
        r.stats = _arena(x.device).take(2 * sizes[stats_label] * _lib.STAT_REPLICAS)   # replicated f64 channel sums of y
    d = _lib.ContractDesc()
    d.A, d.X, d.Y, d.tab = a.data_ptr(), x.data_ptr(), y.data_ptr(), p.tables.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.stats = r.stats.data_ptr() if r.stats is not None else None
    d.G, d.M, d.N, d.K, d.splitk, d.a_kfast, d.x_kfast = p.G, p.M, p.N, p.K, p.splitk, p.a_kfast, p.x_kfast
    d.x_vec = 1 if (p.x_vec and x.data_ptr() % 16 == 0) else 0
    d.stat_ch = sizes[stats_label] if stats_label is not None else 0
    d.mode, d.ws, d.chain = p.mode, None, 0          # the launcher points ws at zeroed scratch / links chains
    r.desc = d
    return r


_MAX_CONTRACT_BATCH = 16
_ACC_MAX_FLOATS = 1 << 22      # shared-input gradients: fp32 atomics into one buffer up to this size, else separate outputs + one sum


def _contract_launch(builders, device, groups=None, chains=None):
    """builders: callables out -> _Prep (out = pre-zeroed flat buffer the result must be written into, or None).
    Runs all contractions in as few launches as possible.  Split-K outputs are carved from ONE zeroed buffer;
    `groups[i]` (optional) names an accumulation group: all members add (fp32 atomics) into one shared zeroed output;
    `chains` (optional): lists of builder indices whose results are to be SUMMED - when all of them take the streaming
    kernel with the same output geometry they run as one chained problem (the first one's `y` holds the sum, the others
    get `chained = True` and no output), otherwise nothing changes and the caller sums."""
    probe = [b(None) for b in builders]            # plan lookup is cached, so probing is cheap
    groups = groups or [None] * len(builders)
    slots, order = {}, []                          # slot key -> numel
    for i, r in enumerate(probe):
        key = ("g", groups[i]) if groups[i] is not None else (("s", i) if r.zero else None)
        if key is not None and key not in slots:
            slots[key] = r.y.numel()
            order.append(key)
        if r.ws_floats:
            slots[("w", i)] = r.ws_floats
            order.append(("w", i))
    if order:
        offs, total = {}, 0
        for key in order:
            offs[key] = total
            total += (slots[key] + 3) & ~3
        zbuf, _ = _zeros(total, device)
        for i, r in enumerate(probe):
            key = ("g", groups[i]) if groups[i] is not None else (("s", i) if r.zero else None)
            if key is not None:
                stats = r.stats
                probe[i] = builders[i](zbuf[offs[key]:offs[key] + slots[key]])
                probe[i].stats = stats
                if groups[i] is not None:
                    probe[i].desc.accumulate = 1
            if r.ws_floats:
                probe[i].desc.ws = zbuf[offs[("w", i)]:].data_ptr()
    for ch in (chains or ()):
        ds = [probe[i].desc for i in ch]
        same_chunk = len({i // _MAX_CONTRACT_BATCH for i in ch}) == 1
        if (len(ch) > 1 and same_chunk and all(d.mode == 1 and not d.accumulate and not d.stats and not d.bias for d in ds)
                and len({(d.G, d.M, d.N) for d in ds}) == 1):
            for a, b in zip(ch[:-1], ch[1:]):
                probe[a].desc.chain = (b % _MAX_CONTRACT_BATCH) + 1
            for i in ch[1:]:
                probe[i].chained, probe[i].y = True, None
    for c0 in range(0, len(probe), _MAX_CONTRACT_BATCH):
        chunk = probe[c0:c0 + _MAX_CONTRACT_BATCH]
        arr = (_lib.ContractDesc * len(chunk))(*[r.desc for r in chunk])
        _lib.call("cg_contract_many", arr, len(chunk), _stream(next(r.y for r in chunk if r.y is not None)))
    return probe


def _contract_raw(spec, a, x, bias=None, bias_label=None, sizes=None, sa=None, sx=None, oa=0, ox=0, stats_label=None):
    """Y = contraction of a and x per einsum-style `spec`; returns a fresh contiguous tensor.
    `sa`/`sx` (label -> element stride) and `oa`/`ox` override the operands' own strides, which is
    how dilated / halo-shifted windows are addressed without materialising them."""
    r = _contract_launch([lambda out: _contract_prepare(spec, a, x, bias, bias_label, sizes, sa, sx, oa, ox, stats_label, out)],
                         x.device)[0]
    return (r.y, r.stats) if stats_label is not None else r.y


_ones = {}


def _sum_keep(t, labels, keep):
    """Sum a tensor over every index except those in `keep` (contraction with a broadcast one)."""
    one = _ones.get(t.device)
    if one is None:
        one = _ones[t.device] = torch.ones(1, dtype=torch.float32, device=t.device)
    rest = "".join(l for l in labels if l not in keep)
    if not rest:
        return t
    if t.dim() == 4 and keep == labels[1]:
        # bias gradient of a convolution: a plain channel sum (the contraction took 170 us for a 5.6 MB tensor: split-K with
        # one output column)
        out, _ = _zeros(t.shape[1], t.device)
        v = _view4(t)
        _lib.call("cg_chan_sum", _ptr(t), ctypes.byref(v), _ptr(out), _stream(t))
        return out
    sizes = {l: n for l, n in zip(labels, t.shape)}
    return _contract_raw("%s,%s->%s" % (labels, rest, keep), t, one, sizes=sizes, sx={l: 0 for l in rest})


def _sum_keep_builder(t, labels, keep):
    one = _ones.get(t.device)
    if one is None:
        one = _ones[t.device] = torch.ones(1, dtype=torch.float32, device=t.device)
    rest = "".join(l for l in labels if l not in keep)
    sizes = {l: n for l, n in zip(labels, t.shape)}
    return lambda out: _contract_prepare("%s,%s->%s" % (labels, rest, keep), t, one, sizes=sizes, sx={l: 0 for l in rest}, out=out)


class _ContractMany(torch.autograd.Function):
    """Several independent contractions in one launch (forward) and all their gradients in one launch (backward)."""

    @staticmethod
    def forward(ctx, metas, *ts):
        ctx.set_materialize_grads(False)      # the channel-sum outputs never receive a gradient: do not zero-fill one
        n = len(metas)
        trip = [ts[3 * i:3 * i + 3] for i in range(n)]
        builders = [(lambda out, m=m, t=t: _contract_prepare(m[0], t[0], t[1], t[2], m[1], stats_label=m[2], out=out))
                    for m, t in zip(metas, trip)]
        preps = _contract_launch(builders, trip[0][1].device)
        ctx.metas = metas
        ctx.save_for_backward(*[t for tr in trip for t in tr[:2]])
        outs = []
        for r in preps:
            if r.stats is not None:
                ctx.mark_non_differentiable(r.stats)
            outs += [r.y, r.stats]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        saved = ctx.saved_tensors
        builders, slots, groups = [], [], []
        shared = {}      # inputs that feed several problems of the stage: their gradients accumulate in one buffer
        for i in range(len(ctx.metas)):
            x = saved[2 * i + 1]
            if ctx.needs_input_grad[2 + 3 * i] and grads[2 * i] is not None:
                shared.setdefault((x.data_ptr(), tuple(x.shape), tuple(x.stride())), []).append(i)
        dev = saved[0].device
        # order: weight, input, bias gradient per map (measured: input gradients first, so that all of a shared input's
        # land in one launch chunk, is slower at both B=16 and B=256)
        for i, (spec, bias_label, _) in enumerate(ctx.metas):
            a, x, dy = saved[2 * i], saved[2 * i + 1], grads[2 * i]
            if dy is None:                    # this output was not used downstream
                continue
            ins, ly = spec.split("->")
            la, lx = ins.split(",")
            if ctx.needs_input_grad[1 + 3 * i]:
                builders.append(lambda out, s="%s,%s->%s" % (ly, lx, la), dy=dy, x=x: _contract_prepare(s, dy, x, out=out))
                slots.append(3 * i); groups.append(None)
            if ctx.needs_input_grad[2 + 3 * i]:
                key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()))
                builders.append(lambda out, s="%s,%s->%s" % (la, ly, lx), a=a, dy=dy: _contract_prepare(s, a, dy, out=out))
                # large tensors: float atomics are slower than separate outputs + adds (MI355X ~1.3 TB/s of atomic bytes)
                small = x.numel() * len(shared[key]) <= _ACC_MAX_FLOATS
                slots.append(3 * i + 1); groups.append(key if (len(shared[key]) > 1 and small) else None)
            if ctx.needs_input_grad[3 + 3 * i]:
                builders.append(_sum_keep_builder(dy, ly, bias_label))
                slots.append(3 * i + 2); groups.append(None)
        res = [None] * (3 * len(ctx.metas))
        if builders:
            seen = set()
            big = {}            # shared input too large for atomics: its gradients are separate outputs, summed by ONE launch below
            chains = {}         # ... unless they can run as one chained streaming problem (no intermediate tensors at all)
            for bi, (slot, grp) in enumerate(zip(slots, groups)):
                if grp is None and slot % 3 == 1:
                    x = saved[2 * (slot // 3) + 1]
                    key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()))
                    if len(shared.get(key, ())) > 1:
                        chains.setdefault(key, []).append(bi)
            for slot, grp, r in zip(slots, groups, _contract_launch(builders, dev, groups, _chain_lists(chains.values(), builders))):
                if grp is not None:
                    if grp in seen:
                        continue          # the shared buffer already holds the sum; hand it to autograd once
                    seen.add(grp)
                elif r.chained:
                    continue              # summed into the head of its chain in registers
                elif slot % 3 == 1:
                    x = saved[2 * (slot // 3) + 1]
                    key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()))
                    if len(shared.get(key, ())) > 1:
                        big.setdefault(key, []).append((slot, r.y))
                        continue
                res[slot] = r.y
            for parts in big.values():      # autograd would add them pairwise: (n-1) x (2 reads + 1 write) of the tensor
                res[parts[0][0]] = _sum_tensors([y for _, y in parts])
        return (None,) + tuple(res)


_CHAIN = bool(int(__import__("os").environ.get("CISTGCN_CHAIN", "1")))        # tuning aid: 0 = separate outputs + one sum


def _chain_lists(cands, builders):
    """Split each candidate list (builder indices whose outputs are to be summed) into runs that share the streaming
    geometry; runs of one stay unchained."""
    out = []
    if not _CHAIN:
        return out
    for idxs in cands:
        byg = {}
        for i in idxs:
            d = builders[i](None).desc            # cached plan: cheap
            if d.mode == 1:
                byg.setdefault((d.G, d.M, d.N, i // _MAX_CONTRACT_BATCH), []).append(i)
        out += [v for v in byg.values() if len(v) > 1]
    return out


def contract_many(items):
    """items: (spec, a, x, bias, bias_label, stats_label) tuples -> list of (y, channel sums or None).
    All contractions run in one launch; so do all their gradients."""
    metas = tuple((it[0], it[4], it[5]) for it in items)
    flat = []
    for it in items:
        flat += [it[1], it[2], it[3]]
    out = _ContractMany.apply(metas, *flat)
    return [(out[2 * i], out[2 * i + 1]) for i in range(len(items))]


class _Contract(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, x, bias, spec, bias_label, stats_label):
        ctx.set_materialize_grads(False)      # the channel-sum output never receives a gradient: do not zero-fill one
        ctx.spec, ctx.bias_label = spec, bias_label
        ctx.save_for_backward(a, x)
        if stats_label is None:
            return _contract_raw(spec, a, x, bias, bias_label), None
        y, stats = _contract_raw(spec, a, x, bias, bias_label, stats_label=stats_label)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _=None):
        if dy is None:
            return None, None, None, None, None, None
        a, x = ctx.saved_tensors
        ins, ly = ctx.spec.split("->")
        la, lx = ins.split(",")
        da = dx = db = None
        if ctx.needs_input_grad[0]:
            da = _contract_raw("%s,%s->%s" % (ly, lx, la), dy, x)
        if ctx.needs_input_grad[1]:
            dx = _contract_raw("%s,%s->%s" % (la, ly, lx), a, dy)
        if ctx.needs_input_grad[2]:
            db = _sum_keep(dy, ly, ctx.bias_label)
        return da, dx, db, None, None, None


def contract(spec, a, x, bias=None, bias_label=None):
    """einsum-style contraction `spec` = "<a idx>,<x idx>-><y idx>", optional bias along one index."""
    return _Contract.apply(a, x, bias, spec, bias_label, None)[0]


def contract_stats(spec, a, x, bias=None, bias_label=None, stats_label="o"):
    """contraction + f64 per-channel (sum, sum of squares) of the result along `stats_label` from the kernel's
    epilogue; stats is None when the plan needs split-K (the caller then reduces separately)."""
    return _Contract.apply(a, x, bias, spec, bias_label, stats_label)


# ----------------------------------------------------------------------------------------------
# fused BatchNorm / Dropout / residual / PReLU row kernel
# ----------------------------------------------------------------------------------------------
_ROW_BATCH = 6


def _na_fill_fwd(a, x, pre, add, gamma, beta, alpha, cfg, pending_stats):
    """Fill the forward part of a CgNormAct; returns (y, save).  Problems whose batch statistics are not yet
    available are appended to `pending_stats` (reduced together by cg_chan_stats_many)."""
    _chk(x, "x")
    dev = x.device
    v = _view4(x)
    B, C, P = v.n[0], v.n[1], v.n[2] * v.n[3]
    y = torch.empty(x.shape, dtype=torch.float32, device=dev)
    a.x, a.xv, a.y, a.yv = x.data_ptr(), v, y.data_ptr(), _view4(y)
    if pre is not None:
        if tuple(pre.shape) != (B, C) or not pre.is_contiguous():
            raise ValueError("norm_act: gate must be a contiguous (B,C) tensor")
        a.pre = pre.data_ptr()
    if add is not None:
        if add.shape != x.shape:
            raise ValueError("norm_act: addend shape %s != %s" % (tuple(add.shape), tuple(x.shape)))
        a.add, a.av = add.data_ptr(), _view4(add)
    a.add_post = 1 if cfg.get("add_post") else 0
    bn = cfg.get("bn")
    save = None
    if bn is not None:
        train = cfg["train"]
        a.bn_mode = 1 if train else 2
        a.gamma, a.beta = gamma.data_ptr(), beta.data_ptr()
        a.running_mean, a.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        a.num_batches_tracked = bn.num_batches_tracked.data_ptr()
        a.momentum, a.eps = bn.momentum, bn.eps
        save = torch.empty(2, C, dtype=torch.float32, device=dev)
        a.save_mean, a.save_rstd = save[0].data_ptr(), save[1].data_ptr()
        if train:
            if B * P <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
            stats = cfg.get("stats")
            if stats is None:
                stats = _arena(dev).take(2 * C * _lib.STAT_REPLICAS)
                it = _lib.StatsArgs()
                it.x, it.xv, it.pre, it.stats = x.data_ptr(), v, (pre.data_ptr() if pre is not None else None), stats.data_ptr()
                pending_stats.append(it)
            a.stats = stats.data_ptr()
    p = float(cfg.get("drop_p", 0.0)) if cfg.get("train") else 0.0
    if p > 0.0:
        a.drop_p, a.seed, a.salt = p, seed_state(dev).data_ptr(), cfg["salt"]
    if alpha is not None:
        a.alpha, a.alpha_n = alpha.data_ptr(), alpha.numel()
    emitted = None
    if cfg.get("emit_stats"):
        emitted = _arena(dev).take(2 * C * _lib.STAT_REPLICAS)
        a.ystats = emitted.data_ptr()
    return y, save, p, emitted


class _NormActMany(torch.autograd.Function):
    """Up to any number of independent fused BatchNorm/Dropout/residual/PReLU row problems; forward is one
    statistics launch (only for inputs without precomputed sums) + one launch per 6 problems, backward one
    reduction launch + one apply launch per 6 problems."""

    @staticmethod
    def forward(ctx, cfgs, *ts):
        ctx.set_materialize_grads(False)      # emitted channel sums never receive a gradient
        n = len(cfgs)
        six = [ts[6 * i:6 * i + 6] for i in range(n)]
        arr = (NormAct * n)()
        pending, ys, saves, ps, emitted = [], [], [], [], []
        for i in range(n):
            y, save, p, em = _na_fill_fwd(arr[i], *six[i], cfgs[i], pending)
            ys.append(y); saves.append(save); ps.append(p); emitted.append(em)
        stream = _stream(six[0][0])
        for c0 in range(0, len(pending), _ROW_BATCH):
            chunk = pending[c0:c0 + _ROW_BATCH]
            _lib.call("cg_chan_stats_many", (_lib.StatsArgs * len(chunk))(*chunk), len(chunk), stream)
        for c0 in range(0, n, _ROW_BATCH):
            m = min(_ROW_BATCH, n - c0)
            _lib.call("cg_norm_act_fwd_many", ctypes.cast(ctypes.byref(arr[c0]), ctypes.POINTER(NormAct)), m, stream)
        ctx.cfgs, ctx.ps, ctx.modes = cfgs, ps, [arr[i].bn_mode for i in range(n)]
        ctx.seed_epoch = seed_epoch(six[0][0].device) if any(p > 0.0 for p in ps) else None
        flat = []
        for i in range(n):
            flat += list(six[i]) + [saves[i]]
        ctx.save_for_backward(*flat)
        for em in emitted:
            if em is not None:
                ctx.mark_non_differentiable(em)
        return tuple(ys) + tuple(emitted)

    @staticmethod
    def backward(ctx, *dys):
        n = len(ctx.cfgs)
        saved = ctx.saved_tensors
        _check_seed_epoch(ctx, saved[0].device, "norm_act")
        arr = (NormAct * n)()
        flags = (ctypes.c_int * n)()
        grads = []
        live = [i for i in range(n) if dys[i] is not None]
        if len(live) != n:
            raise RuntimeError("norm_act_many: an output was not used in the loss; call the problems separately")
        for i in range(n):
            x, pre, add, gamma, beta, alpha, save = saved[7 * i:7 * i + 7]
            cfg, dy, a = ctx.cfgs[i], dys[i], arr[i]
            dev = x.device
            v = _view4(x)
            B, C = v.n[0], v.n[1]
            a.x, a.xv = x.data_ptr(), v
            if pre is not None:
                a.pre = pre.data_ptr()
            add_post = 1 if cfg.get("add_post") else 0
            a.add_post = add_post
            if add is not None:
                a.add, a.av = add.data_ptr(), _view4(add)
            a.bn_mode = ctx.modes[i]
            bn = cfg.get("bn")
            if bn is not None:
                a.gamma, a.beta = gamma.data_ptr(), beta.data_ptr()
                a.running_mean, a.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                a.momentum, a.eps = bn.momentum, bn.eps
                a.save_mean, a.save_rstd = save[0].data_ptr(), save[1].data_ptr()
            if ctx.ps[i] > 0.0:
                a.drop_p, a.seed, a.salt = ctx.ps[i], seed_state(dev).data_ptr(), cfg["salt"]
            nalpha = 0
            if alpha is not None:
                nalpha = alpha.numel()
                a.alpha, a.alpha_n = alpha.data_ptr(), nalpha
            a.dy, a.dyv = dy.data_ptr(), _view4(dy)
            need = ctx.needs_input_grad[1 + 6 * i:7 + 6 * i]
            dx = dpre = dadd = dgamma = dbeta = dalpha = None
            if need[0]:
                dx = torch.empty(x.shape, dtype=torch.float32, device=dev)
                a.dx, a.dxv = dx.data_ptr(), _view4(dx)
            if pre is not None and need[1]:
                dpre = torch.empty(B, C, dtype=torch.float32, device=dev)
                a.dpre = dpre.data_ptr()
            if add is not None and need[2]:
                if add_post:
                    dadd = dy
                else:
                    dadd = torch.empty(add.shape, dtype=torch.float32, device=dev)
                    a.dadd, a.dav = dadd.data_ptr(), _view4(dadd)
            flags[i] = 1 if (bn is not None or alpha is not None) else 0
            if flags[i]:
                a.red = _arena(dev).take(2 * C + (_lib.ALPHA_SLOTS if nalpha == 1 else nalpha)).data_ptr()
                if bn is not None:
                    dgamma = torch.empty(C, dtype=torch.float32, device=dev)
                    dbeta = torch.empty(C, dtype=torch.float32, device=dev)
                    a.dgamma, a.dbeta = dgamma.data_ptr(), dbeta.data_ptr()
                if alpha is not None:
                    dalpha = torch.empty(alpha.shape, dtype=torch.float32, device=dev)
                    a.dalpha = dalpha.data_ptr()
            grads += [dx, dpre, dadd, dgamma, dbeta, dalpha]
        stream = _stream(saved[0])
        for c0 in range(0, n, _ROW_BATCH):
            m = min(_ROW_BATCH, n - c0)
            _lib.call("cg_norm_act_bwd_many", ctypes.cast(ctypes.byref(arr[c0]), ctypes.POINTER(NormAct)),
                      ctypes.cast(ctypes.byref(flags, 4 * c0), ctypes.POINTER(ctypes.c_int)), m, stream)
        return (None,) + tuple(grads)


def _na_args(x, bn=None, train=False, pre=None, add=None, add_post=False, drop_p=0.0, salt=0, prelu=None, stats=None,
             emit_stats=False):
    if isinstance(x, tuple):          # (tensor, channel sums) as returned by the contraction helpers
        x, st = x
        stats = st if st is not None else stats
    cfg = {"bn": bn, "train": bool(train), "add_post": add_post, "drop_p": drop_p, "salt": salt, "stats": stats,
           "emit_stats": bool(emit_stats)}
    return cfg, (x, pre, add, bn.weight if bn is not None else None, bn.bias if bn is not None else None,
                 prelu.weight if prelu is not None else None)


def norm_act_many(calls):
    """calls: list of keyword dicts as accepted by `norm_act`; all problems run in shared launches."""
    cfgs, flat = [], []
    for kw in calls:
        cfg, six = _na_args(**kw)
        cfgs.append(cfg)
        flat += list(six)
    out = _NormActMany.apply(tuple(cfgs), *flat)
    n = len(cfgs)
    return [(out[i], out[n + i]) if cfgs[i]["emit_stats"] else out[i] for i in range(n)]


def norm_act(x, bn=None, train=False, pre=None, add=None, add_post=False, drop_p=0.0, salt=0, prelu=None, stats=None,
             emit_stats=False):
    """y = PReLU(Dropout(BN(x * pre)) [+ add]) [+ add];  every stage optional.
    `bn` is the parameter holder (an nn.BatchNorm*), `prelu` an nn.PReLU; `stats` are precomputed
    f64 channel sums of x (the contraction / fused ST-GCN kernels emit them); `emit_stats` returns
    (y, f64 channel sums of y) for a BatchNorm that consumes y next."""
    return norm_act_many([dict(x=x, bn=bn, train=train, pre=pre, add=add, add_post=add_post, drop_p=drop_p, salt=salt,
                               prelu=prelu, stats=stats, emit_stats=emit_stats)])[0]


# ----------------------------------------------------------------------------------------------
# reductions over positions, SE gate
# ----------------------------------------------------------------------------------------------
class _ReduceBC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        _chk(x)
        v = _view4(x)
        out = torch.empty(v.n[0], v.n[1], dtype=torch.float32, device=x.device)
        arg = torch.empty(v.n[0], v.n[1], dtype=torch.int32, device=x.device) if kind == 1 else None
        _lib.call("cg_reduce_bc", _ptr(x), ctypes.byref(v), kind, _ptr(out), _ptr(arg), _stream(x))
        ctx.kind, ctx.shape = kind, x.shape
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, dout):
        arg, = ctx.saved_tensors
        dout = dout if dout.is_contiguous() else _copy(dout)
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dout.device)
        v = _view4(dx)
        _lib.call("cg_reduce_bc_bwd", _ptr(dout), _ptr(arg), ctx.kind, _ptr(dx), ctypes.byref(v), _stream(dout))
        return dx, None


def mean_bc(x):
    """(B,C,...) -> (B,C) mean over everything behind the channel axis."""
    return _ReduceBC.apply(x, 0)


def max_bc(x):
    """(B,C,...) -> (B,C) max over everything behind the channel axis (gradient to the first arg-max)."""
    return _ReduceBC.apply(x, 1)


class _SEGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pooled, w1, w2):
        B, C = pooled.shape
        H = w1.shape[0]
        gate = torch.empty(B, C, dtype=torch.float32, device=pooled.device)
        _lib.call("cg_se_gate_fwd", _ptr(pooled), _ptr(w1), _ptr(w2), _ptr(gate), B, C, H, _stream(pooled))
        ctx.save_for_backward(pooled, w1, w2, gate)
        return gate

    @staticmethod
    def backward(ctx, dgate):
        pooled, w1, w2, gate = ctx.saved_tensors
        B, C = pooled.shape
        H = w1.shape[0]
        dgate = dgate if dgate.is_contiguous() else _copy(dgate)
        dp = torch.empty_like(pooled)
        dw, _ = _zeros(2 * C * H, pooled.device)
        dw1, dw2 = dw[:C * H].view(w1.shape), dw[C * H:].view(w2.shape)
        _lib.call("cg_se_gate_bwd", _ptr(pooled), _ptr(w1), _ptr(w2), _ptr(gate), _ptr(dgate), _ptr(dp), _ptr(dw1),
                  _ptr(dw2), B, C, H, 1, _stream(pooled))
        return dp, dw1, dw2


def se_gate(pooled, w1, w2):
    return _SEGate.apply(pooled, w1, w2)


# ----------------------------------------------------------------------------------------------
# strided copies: cat / pad / sums
# ----------------------------------------------------------------------------------------------
def _add_into(y, a, b=None, c=None):
    vy, va = _view4(y), _view4(a)
    vb = _view4(b) if b is not None else None
    vc = _view4(c) if c is not None else None
    _lib.call("cg_add3", _ptr(y), ctypes.byref(vy), _ptr(a), ctypes.byref(va), _ptr(b),
              ctypes.byref(vb) if vb is not None else None, _ptr(c), ctypes.byref(vc) if vc is not None else None, _stream(y))


def _sum_tensors(ts):
    """sum of same-shape tensors in one launch per eight inputs (cg_sum_many)"""
    if len(ts) == 1:
        return ts[0]
    out = torch.empty(ts[0].shape, dtype=torch.float32, device=ts[0].device)
    y = out if out.dim() >= 2 else out.view(1, -1)
    if y.dim() > 4:
        y = y.reshape(y.shape[0], -1)
    ts = list(ts)
    while ts:
        part, ts = ts[:_SUM_MAX], ts[_SUM_MAX:]
        if ts:                                  # more than eight: fold the partial sum into the next round
            ts.insert(0, out)
        arr = (_lib.SumItem * len(part))()
        for i, g in enumerate(part):
            g = g if g.dim() >= 2 else g.view(1, -1)
            if g.dim() > 4:
                g = g.reshape(g.shape[0], -1)
            arr[i].a, arr[i].av = g.data_ptr(), _view4(g)
        vy = _view4(y)
        _lib.call("cg_sum_many", _ptr(y), ctypes.byref(vy), arr, len(part), _stream(y))
    return out


class _Fanout(torch.autograd.Function):
    """n aliases of one tensor, one per consumer: backward sums the consumers' gradients with ONE kernel instead of
    one autograd accumulation kernel per extra consumer."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        return _sum_tensors(gs), None


_SUM_MAX = 8
_FANOUT = bool(int(__import__("os").environ.get("CISTGCN_FANOUT", "1")))      # tuning aid: 0 = let autograd accumulate


def fanout(x, n):
    """n autograd-independent aliases of x (see _Fanout); n <= 1 returns (x,)"""
    if n <= 1 or not _FANOUT or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * max(n, 1)
    return _Fanout.apply(x, n)


def _copy(t):
    y = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    if t.dim() > 4 or t.dim() < 2:
        flat = t.reshape(1, -1) if t.is_contiguous() else None
        if flat is None:
            raise ValueError("copy: unsupported rank")
        _add_into(y.view(1, -1), flat)
    else:
        _add_into(y, t)
    return y


class _Cat(torch.autograd.Function):
    """Concatenation along the channel axis; inputs flagged `bcast` are (B,C) maps broadcast over
    the positions (F.interpolate of a 1x1 pool, CISTGCN.py:76)."""

    @staticmethod
    def forward(ctx, bcast, *ts):
        ref = next(t for t, bc in zip(ts, bcast) if not bc)
        chans = [t.shape[1] for t in ts]
        shape = list(ref.shape)
        shape[1] = sum(chans)
        y = torch.empty(shape, dtype=torch.float32, device=ref.device)
        c0, items = 0, []
        for t, bc, c in zip(ts, bcast, chans):
            dst = y.narrow(1, c0, c)
            src = t.view(t.shape[0], c, *([1] * (ref.dim() - 2))).expand_as(dst) if bc else t
            it = _lib.CopyItem()
            it.y, it.yv, it.a, it.av = dst.data_ptr(), _view4(dst), src.data_ptr(), _view4(src)
            items.append(it)
            c0 += c
        for i0 in range(0, len(items), 4):      # all slices of the concatenation in one launch (four per launch)
            chunk = items[i0:i0 + 4]
            _lib.call("cg_copy_many", (_lib.CopyItem * len(chunk))(*chunk), len(chunk), _stream(y))
        ctx.bcast, ctx.chans = bcast, chans
        return y

    @staticmethod
    def backward(ctx, dy):
        outs, c0 = [], 0
        for i, (bc, c) in enumerate(zip(ctx.bcast, ctx.chans)):
            g = dy.narrow(1, c0, c)
            if not ctx.needs_input_grad[i + 1]:
                g = None
            elif bc:
                v = _view4(g)
                m = torch.empty(g.shape[0], c, dtype=torch.float32, device=dy.device)
                _lib.call("cg_reduce_bc", _ptr(g), ctypes.byref(v), 2, _ptr(m), None, _stream(dy))
                g = m
            outs.append(g)
            c0 += c
        return (None,) + tuple(outs)


def cat_channels(ts, bcast=None):
    bcast = tuple(bcast) if bcast is not None else (False,) * len(ts)
    return _Cat.apply(bcast, *ts)


class _Add3(torch.autograd.Function):
    """a + b + c for 4-D operands; size-1 axes broadcast (the `x[:, -1:]` term of CISTGCN.py:597)."""

    @staticmethod
    def forward(ctx, a, b, c):
        shape = torch.broadcast_shapes(a.shape, b.shape, c.shape if c is not None else a.shape)
        y = torch.empty(shape, dtype=torch.float32, device=a.device)
        _add_into(y, a.expand(shape), b.expand(shape), c.expand(shape) if c is not None else None)
        ctx.shapes = (a.shape, b.shape, c.shape if c is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        outs = []
        labels = "abcd"[:dy.dim()]
        for i, shp in enumerate(ctx.shapes):
            if shp is None or not ctx.needs_input_grad[i]:
                outs.append(None)
                continue
            keep = "".join(l for l, n, m in zip(labels, shp, dy.shape) if n == m)
            g = dy if len(keep) == dy.dim() else _sum_keep(dy, labels, keep).view(shp)
            outs.append(g)
        return tuple(outs)


def add3(a, b, c=None):
    return _Add3.apply(a, b, c)


# ----------------------------------------------------------------------------------------------
# dilated 3x3 convolutions of the FPN time extrapolator (CISTGCN.py:54-68)
# ----------------------------------------------------------------------------------------------
def _halo(x, pad):
    """zero halo of `pad` on the last two axes of a (possibly strided) 4-D tensor"""
    B, C, H, W = x.shape
    xp = _zeros(B * C * (H + 2 * pad) * (W + 2 * pad), x.device)[0].view(B, C, H + 2 * pad, W + 2 * pad)
    _add_into(xp[:, :, pad:pad + H, pad:pad + W], x)
    return xp


class _DilatedConvs(torch.autograd.Function):
    """Three 3x3 convolutions of one input with dilation = padding = 1, 2, 3.  The input is copied
    once into a zero-halo buffer; each convolution is a contraction whose window/dilation lives in
    the offset tables.  Backward-data is the same contraction on the halo-padded output gradient
    with negated window strides."""
    PAD = 3

    @staticmethod
    def forward(ctx, x, dils, *wb):
        B, C, H, W = x.shape
        pad = _DilatedConvs.PAD
        xp = _halo(x, pad)
        sB, sC, sH, sW = xp.stride()
        builders = []
        for i, d in enumerate(dils):
            w, b = wb[2 * i], wb[2 * i + 1]
            sizes = {"b": B, "c": C, "o": w.shape[0], "i": 3, "j": 3, "h": H, "w": W}
            builders.append(lambda out, w=w, b=b, d=d, sizes=sizes: _contract_prepare(
                "ocij,bcijhw->bohw", w, xp, b, "o", sizes=sizes,
                sx={"b": sB, "c": sC, "i": d * sH, "j": d * sW, "h": sH, "w": sW}, ox=(pad - d) * (sH + sW), out=out))
        outs = [r.y for r in _contract_launch(builders, x.device)]      # the three dilations in one launch
        ctx.dils, ctx.shape = dils, x.shape
        ctx.save_for_backward(xp, *wb[0::2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        xp, *ws = ctx.saved_tensors
        B, C, H, W = ctx.shape
        pad = _DilatedConvs.PAD
        sB, sC, sH, sW = xp.stride()
        builders, slots = [], []
        for i, d in enumerate(ctx.dils):
            w, dy = ws[i], dys[i]
            sizes = {"b": B, "c": C, "o": w.shape[0], "i": 3, "j": 3, "h": H, "w": W}
            if ctx.needs_input_grad[2 + 2 * i]:
                builders.append(lambda out, dy=dy, d=d, sizes=sizes: _contract_prepare(
                    "bohw,bcijhw->ocij", dy, xp, sizes=sizes,
                    sx={"b": sB, "c": sC, "i": d * sH, "j": d * sW, "h": sH, "w": sW}, ox=(pad - d) * (sH + sW), out=out))
                slots.append(("w", i))
            if ctx.needs_input_grad[3 + 2 * i]:
                builders.append(_sum_keep_builder(dy, "bohw", "o"))
                slots.append(("b", i))
            if ctx.needs_input_grad[0]:
                dyp = _halo(dy, pad)
                tB, tO, tH, tW = dyp.stride()
                # dx[b,c,h,w] = sum_{o,i,j} w[o,c,i,j] * dy[b,o,h-d(i-1),w-d(j-1)]
                builders.append(lambda out, w=w, dyp=dyp, d=d, sizes=sizes, t=(tB, tO, tH, tW): _contract_prepare(
                    "ocij,boijhw->bchw", w, dyp, sizes=sizes,
                    sx={"b": t[0], "o": t[1], "i": -d * t[2], "j": -d * t[3], "h": t[2], "w": t[3]},
                    ox=(pad + d) * (t[2] + t[3]), out=out))
                slots.append(("x", i))
        grads, dxs = [None] * (2 * len(ctx.dils)), []
        if builders:
            for (kind, i), r in zip(slots, _contract_launch(builders, xp.device)):    # all nine gradients in one launch
                if kind == "w":
                    grads[2 * i] = r.y
                elif kind == "b":
                    grads[2 * i + 1] = r.y
                else:
                    dxs.append(r.y)
        dx = None
        if dxs:
            dx = torch.empty(ctx.shape, dtype=torch.float32, device=xp.device)
            _add_into(dx, dxs[0], dxs[1] if len(dxs) > 1 else None, dxs[2] if len(dxs) > 2 else None)
        return (dx, None) + tuple(grads)


def dilated_convs(x, convs):
    """convs: three nn.Conv2d holders (3x3, dilation = padding = 1,2,3); returns their three outputs."""
    dils = tuple(int(c.dilation[0]) for c in convs)
    for c, d in zip(convs, dils):
        if tuple(c.kernel_size) != (3, 3) or tuple(c.padding) != (d, d) or d > _DilatedConvs.PAD:
            raise ValueError("dilated_convs: unsupported convolution geometry %s" % (c,))
    wb = []
    for c in convs:
        wb += [c.weight, c.bias]
    B, C, H, W = x.shape
    O = convs[0].out_channels
    # (one workgroup per sample: below ~64 samples the chip is mostly idle and the generic contraction is faster)
    if (_FPN_KERNELS and B >= _FPN_MIN_BATCH and x.stride(3) == 1 and all(c.out_channels == O and c.bias is not None for c in convs)
            and _lib.lib().cg_fpn_conv_supported(B, C, O, H, W)):
        return _FpnConvs.apply(x, dils, *wb)
    return _DilatedConvs.apply(x, dils, *wb)


_FPN_KERNELS = __import__("os").environ.get("CISTGCN_FPN_KERNELS", "1") != "0"     # 0: the generic contraction (tuning aid)
_FPN_MIN_BATCH = int(__import__("os").environ.get("CISTGCN_FPN_MIN_BATCH", "64"))


class _FpnConvs(torch.autograd.Function):
    """The dilated convolutions of one FPN block on whole samples in LDS (csrc/fpn_conv.hip); same contract as _DilatedConvs."""

    @staticmethod
    def _block(x, dils, ws):
        B, C, H, W = x.shape
        t = _lib.FpnConv()
        t.B, t.C, t.O, t.H, t.W, t.n = B, C, ws[0].shape[0], H, W, len(dils)
        t.x = x.data_ptr()
        t.xs[0], t.xs[1], t.xs[2] = x.stride(0), x.stride(1), x.stride(2)
        for i, d in enumerate(dils):
            t.dil[i], t.w[i] = d, ws[i].data_ptr()
        return t

    @staticmethod
    def forward(ctx, x, dils, *wb):
        ctx.set_materialize_grads(False)
        _chk(x)
        ws = [w if w.is_contiguous() else _copy(w) for w in wb[0::2]]
        t = _FpnConvs._block(x, dils, ws)
        B, C, H, W = x.shape
        ys = [torch.empty(B, w.shape[0], H, W, dtype=torch.float32, device=x.device) for w in ws]
        for i in range(len(dils)):
            t.bias[i], t.y[i] = wb[2 * i + 1].data_ptr(), ys[i].data_ptr()
        _lib.call("cg_fpn_conv_fwd", ctypes.byref(t), _stream(x))
        ctx.dils = dils
        ctx.save_for_backward(x, *ws)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        x, *ws = ctx.saved_tensors
        n = len(ctx.dils)
        if any(d is None for d in dys):
            raise RuntimeError("dilated_convs: every output needs a gradient")
        dys = [d if d.is_contiguous() else _copy(d) for d in dys]
        t = _FpnConvs._block(x, ctx.dils, ws)
        dev = x.device
        dx = torch.empty(x.shape, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dws = [torch.empty_like(w) for w in ws]
        dbs = [torch.empty(w.shape[0], dtype=torch.float32, device=dev) for w in ws]
        zb = torch.empty(int(_lib.lib().cg_fpn_conv_ws_floats(x.shape[0], x.shape[1], ws[0].shape[0], n)), dtype=torch.float32, device=dev)
        for i in range(n):
            t.dy[i], t.dw[i], t.db[i] = dys[i].data_ptr(), dws[i].data_ptr(), dbs[i].data_ptr()
        t.dx, t.ws = _ptr(dx), zb.data_ptr()
        _lib.call("cg_fpn_conv_bwd", ctypes.byref(t), _stream(x))
        del dys
        grads = []
        for i in range(n):
            grads += [dws[i] if ctx.needs_input_grad[2 + 2 * i] else None, dbs[i] if ctx.needs_input_grad[3 + 2 * i] else None]
        return (dx, None) + tuple(grads)


# ----------------------------------------------------------------------------------------------
# stage kernels
# ----------------------------------------------------------------------------------------------
class _FeatureLift(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        if x.dim() != 4 or x.shape[-1] != 3:
            raise ValueError("expected poses of shape (B, T, V, 3), got %s" % (tuple(x.shape),))
        x = x if x.is_contiguous() else _copy(x)
        B, T, V, _ = x.shape
        f = torch.empty(B, 10, T, V, dtype=torch.float32, device=x.device)
        _lib.call("cg_feature_lift_fwd", _ptr(x), _ptr(f), B, T, V, _stream(x))
        ctx.save_for_backward(x)
        return f

    @staticmethod
    def backward(ctx, df):
        x, = ctx.saved_tensors
        B, T, V, _ = x.shape
        df = df if df.is_contiguous() else _copy(df)
        dx = torch.empty_like(x)
        _lib.call("cg_feature_lift_bwd", _ptr(x), _ptr(df), _ptr(dx), B, T, V, _stream(x))
        return dx


def feature_lift(x):
    return _FeatureLift.apply(x)


class _DstdStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if not x.is_contiguous():
            raise ValueError("dstd_stats expects a contiguous (B,C,T,V) tensor")
        B, C, T, V = x.shape
        out = torch.empty(B, 2 + 2 * T, dtype=torch.float32, device=x.device)
        _lib.call("cg_dstd_stats_fwd", _ptr(x), _ptr(out), B, C, T, V, _stream(x))
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, = ctx.saved_tensors
        B, C, T, V = x.shape
        dout = dout if dout.is_contiguous() else _copy(dout)
        dx = torch.empty_like(x)
        _lib.call("cg_dstd_stats_bwd", _ptr(x), _ptr(dout), _ptr(dx), B, C, T, V, _stream(x))
        return dx


def dstd_stats(x):
    return _DstdStats.apply(x)


class _Cumsum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        vx, vy = _view4(x), _view4(y)
        _lib.call("cg_cumsum", _ptr(x), ctypes.byref(vx), _ptr(y), ctypes.byref(vy), 0, _stream(x))
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
        vy, vx = _view4(dy), _view4(dx)
        _lib.call("cg_cumsum", _ptr(dy), ctypes.byref(vy), _ptr(dx), ctypes.byref(vx), 1, _stream(dy))
        return dx


class _Rank1Adj(torch.autograd.Function):
    """Rank-1 adjacency seeds of the Map2Adj towers of one block (CISTGCN.py:183-189), all domains in one launch:
    inputs (s_0, q_0, s_1, q_1, ...) with s (B,V,T), q (B,T,V); outputs o_i (B,V,T,T) for domain 0, (B,T,V,V) for 1."""

    @staticmethod
    def forward(ctx, domains, *sq):
        ctx.set_materialize_grads(False)
        n = len(domains)
        B, V, T = sq[0].shape
        sq = [t if t.is_contiguous() else _copy(t) for t in sq]
        outs = []
        arr = (_lib.Rank1 * n)()
        for i, dom in enumerate(domains):
            s, q = sq[2 * i], sq[2 * i + 1]
            _chk(s, "s"), _chk(q, "q")
            if tuple(s.shape) != (B, V, T) or tuple(q.shape) != (B, T, V):
                raise ValueError("rank1_adj: expected s (B,V,T) and q (B,T,V), got %s / %s" % (tuple(s.shape), tuple(q.shape)))
            o = torch.empty((B, V, T, T) if dom == 0 else (B, T, V, V), dtype=torch.float32, device=s.device)
            arr[i].s, arr[i].q, arr[i].o, arr[i].domain = s.data_ptr(), q.data_ptr(), o.data_ptr(), dom
            outs.append(o)
        _lib.call("cg_rank1_adj_fwd", arr, n, B, T, V, _stream(sq[0]))
        ctx.domains, ctx.dims = domains, (B, T, V)
        ctx.save_for_backward(*sq)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        sq = ctx.saved_tensors
        B, T, V = ctx.dims
        live = [i for i, d in enumerate(douts) if d is not None]
        grads = [None] * (2 * len(ctx.domains))
        if live:
            arr = (_lib.Rank1 * len(live))()
            keep = []
            for k, i in enumerate(live):
                d = douts[i] if (douts[i].is_contiguous() and douts[i].data_ptr() % 16 == 0) else _copy(douts[i])
                ds, dq = torch.empty_like(sq[2 * i]), torch.empty_like(sq[2 * i + 1])
                arr[k].s, arr[k].q, arr[k].dout = sq[2 * i].data_ptr(), sq[2 * i + 1].data_ptr(), d.data_ptr()
                arr[k].ds, arr[k].dq, arr[k].domain = ds.data_ptr(), dq.data_ptr(), ctx.domains[i]
                grads[2 * i], grads[2 * i + 1] = ds, dq
                keep.append(d)
            _lib.call("cg_rank1_adj_bwd", arr, len(live), B, T, V, _stream(sq[0]))
        return (None,) + tuple(grads)


def rank1_adj(pairs):
    """pairs: [(domain, s (B,V,T), q (B,T,V)), ...] (at most two) -> list of adjacency seeds, one launch"""
    flat = []
    for _, s, q in pairs:
        flat += [s, q]
    return list(_Rank1Adj.apply(tuple(int(p[0]) for p in pairs), *flat))


# ----------------------------------------------------------------------------------------------
# evaluation harness counterpart (environment/test.py:97-132): no autograd, the reference runs it under no_grad
# ----------------------------------------------------------------------------------------------
_index_cache = {}


def _index_tensor(values, device):
    key = (tuple(int(v) for v in values), str(_dev(device)))
    t = _index_cache.get(key)
    if t is None:
        t = _index_cache[key] = torch.tensor(key[0], dtype=torch.int32, device=device)
    return t


def gather_joints(x, dim_used):
    """`inputs[:, :, dim_used]` (test.py:101): (B,T,J,3) -> (B,T,len(dim_used),3)"""
    _chk(x, "x")
    if x.dim() != 4 or x.shape[3] != 3:
        raise ValueError("gather_joints: expected (B,T,J,3), got %s" % (tuple(x.shape),))
    dim_used = [int(v) for v in dim_used]
    if not dim_used or min(dim_used) < 0 or max(dim_used) >= x.shape[2]:
        raise IndexError("gather_joints: joint index out of range")
    x = x if x.is_contiguous() else _copy(x)
    B, T, J, _ = x.shape
    y = torch.empty(B, T, len(dim_used), 3, dtype=torch.float32, device=x.device)
    _lib.call("cg_gather_joints", _ptr(x), _ptr(y), _ptr(_index_tensor(dim_used, x.device)), B * T, J, len(dim_used), _stream(x))
    return y


def eval_scatter_mpjpe(pred, target, dim_used, dim_repeat_32=(), dim_repeat_22=()):
    """Post-processing of `_predict` (test.py:121-127) and the per-frame MPJPE (losses.py:50-61, reduce_axis (0,2)) in one
    launch: returns (full-skeleton prediction (B,To,J,3), error per frame (To,))."""
    _chk(pred, "pred"), _chk(target, "target")
    if pred.dim() != 4 or target.dim() != 4 or pred.shape[:2] != target.shape[:2] or pred.shape[3] != 3 or target.shape[3] != 3:
        raise ValueError("eval_scatter_mpjpe: expected pred (B,To,J22,3) and target (B,To,J32,3)")
    B, To, J22, _ = pred.shape
    J32 = target.shape[2]
    if len(dim_used) != J22 or len(dim_repeat_32) != len(dim_repeat_22):
        raise ValueError("eval_scatter_mpjpe: index lists do not match the tensors")
    src = [-1] * J32
    for k, j in enumerate(dim_used):
        src[int(j)] = k
    for j, k in zip(dim_repeat_32, dim_repeat_22):            # applied after the scatter, as in the reference
        src[int(j)] = int(k)
    if max(src) >= J22:
        raise IndexError("eval_scatter_mpjpe: repeated joint index out of range")
    pred = pred if pred.is_contiguous() else _copy(pred)
    target = target if target.is_contiguous() else _copy(target)
    out = torch.empty_like(target)
    err = torch.empty(To, dtype=torch.float32, device=pred.device)
    _lib.call("cg_eval_scatter_mpjpe", _ptr(pred), _ptr(target), _ptr(out), _ptr(err), _ptr(_index_tensor(src, pred.device)),
              B, To, J32, J22, _stream(pred))
    return out, err


def cumsum_time(x):
    """cumulative sum over axis 1 of a 4-D (possibly strided) tensor"""
    return _Cumsum.apply(x)


class _Mpjpe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        _chk(pred), _chk(target)
        if pred.shape != target.shape or pred.shape[-1] != 3:
            raise AssertionError("mpjpe: predicted %s vs target %s" % (tuple(pred.shape), tuple(target.shape)))
        pred = pred if pred.is_contiguous() else _copy(pred)
        target = target if target.is_contiguous() else _copy(target)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        _lib.call("cg_mpjpe_fwd", _ptr(pred), _ptr(target), _ptr(loss), pred.numel() // 3, _stream(pred))
        ctx.save_for_backward(pred, target)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        g = g.reshape(1)
        d = torch.empty_like(pred)
        _lib.call("cg_mpjpe_bwd", _ptr(pred), _ptr(target), _ptr(g), _ptr(d), pred.numel() // 3, _stream(pred))
        return d, None


def mpjpe(pred, target):
    """Mean per-joint position error with full-mean reduction (losses/losses.py:50-61, reduce_axis=[])."""
    return _Mpjpe.apply(pred, target)


# ----------------------------------------------------------------------------------------------
# fused ST-GCN stage
# ----------------------------------------------------------------------------------------------
class _StgcnDomain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, w, bias, domain, want_stats):
        ctx.set_materialize_grads(False)
        if not (x.is_contiguous() and adj.is_contiguous() and w.is_contiguous()):
            raise ValueError("stgcn_domain expects contiguous x (B,C,T,V), Adj and W")
        B, Cin, T, V = x.shape
        Cout = w.shape[0]
        exp = (B, V, T, T) if domain == 0 else (B, T, V, V)
        if tuple(adj.shape) != exp or w.numel() != Cout * Cin:
            raise ValueError("stgcn_domain: adjacency %s / weight %s do not match x %s" % (tuple(adj.shape), tuple(w.shape), tuple(x.shape)))
        y = torch.empty(B, Cout, T, V, dtype=torch.float32, device=x.device)
        stats = _arena(x.device).take(2 * Cout * _lib.STAT_REPLICAS) if want_stats else None
        _lib.call("cg_stgcn_domain_fwd", _ptr(x), _ptr(adj), _ptr(w), _ptr(bias), _ptr(y), _ptr(stats),
                  B, Cin, Cout, T, V, domain, _stream(x))
        ctx.domain = domain
        ctx.save_for_backward(x, adj, w)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, dy, _):
        x, adj, w = ctx.saved_tensors
        B, Cin, T, V = x.shape
        Cout = w.shape[0]
        dy = dy if dy.is_contiguous() else _copy(dy)
        dx = torch.empty_like(x)
        dadj = torch.empty_like(adj)
        dw = torch.empty_like(w)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
        ws, _ = _zeros(_lib.lib().cg_stgcn_domain_bwd_ws_floats(Cin, Cout), x.device)
        _lib.call("cg_stgcn_domain_bwd", _ptr(x), _ptr(adj), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dadj), _ptr(dw), _ptr(db),
                  _ptr(ws), B, Cin, Cout, T, V, ctx.domain, 1, _stream(x))
        return dx, dadj, dw, db, None, None


def stgcn_domain(x, adj, w, bias, domain, want_stats=False):
    """Fused graph product + channel mix (CISTGCN.py:265-266).  domain 0 = "space" (Adj (B,V,T,T)),
    1 = "time" (Adj (B,T,V,V)).  Returns (y, f64 channel sums of y or None)."""
    return _StgcnDomain.apply(x, adj, w, bias, domain, want_stats)


# ----------------------------------------------------------------------------------------------
# tail of a DSTD_GC block as phase kernels (csrc/dstd_tail.hip)
# ----------------------------------------------------------------------------------------------
def _tail_bn(dst, bn, stats, save, train):
    dst.stats = stats.data_ptr() if stats is not None else None
    dst.gamma, dst.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
    dst.running_mean, dst.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    dst.num_batches_tracked = bn.num_batches_tracked.data_ptr() if train else None
    dst.momentum, dst.eps = bn.momentum, bn.eps
    dst.save = save.data_ptr()


class _DstdTail(torch.autograd.Function):
    """y_i (tcn outputs) -> block output, CISTGCN.py:266-269, :388, :305-309, :390 (see csrc/dstd_tail.hip).
    Tensor inputs: y1 y2 r1 r2 w1 w2 | g_t1 b_t1 a_d1 g_t2 b_t2 a_d2 | g_p1 b_p1 a_p1 g_p2 b_p2 a_p2 | Wc g_c b_c a_c |
    se_w1 se_w2 | bres."""

    @staticmethod
    def forward(ctx, cfg, *ts):
        ctx.set_materialize_grads(False)
        (y1, y2, r1, r2, w1, w2, gt1, bt1, ad1, gt2, bt2, ad2, gp1, bp1, ap1, gp2, bp2, ap2, wc, gc, bc, ac, sw1, sw2, bres) = ts
        for t in (y1, y2, r1, r2, w1, w2, bres):
            _chk(t)
            if not t.is_contiguous():
                raise ValueError("dstd_tail expects contiguous tensors")
        B, C, T, V = y1.shape
        dev, train = y1.device, bool(cfg["train"])
        f32 = torch.float32
        t = _lib.DstdTail()
        t.B, t.C, t.T, t.V, t.train = B, C, T, V, 1 if train else 0
        saves = torch.empty(5, 2, C, dtype=f32, device=dev)           # tcn x2, prelu1/2, compressor: mean / rstd
        t.y[0], t.y[1], t.r[0], t.r[1], t.w[0], t.w[1] = (v.data_ptr() for v in (y1, y2, r1, r2, w1, w2))
        arena = _arena(dev)
        zst = [arena.take(2 * C * _lib.STAT_REPLICAS) for _ in range(2)] if train else [None, None]
        hst = arena.take(2 * C * _lib.STAT_REPLICAS) if train else None
        bns = cfg["bn"]                                                # (tcn1, tcn2, prelu1, prelu2, compressor) parameter holders
        _tail_bn(t.bn_t[0], bns[0], cfg["ystats"][0], saves[0], train); _tail_bn(t.bn_t[1], bns[1], cfg["ystats"][1], saves[1], train)
        _tail_bn(t.bn_p[0], bns[2], zst[0], saves[2], train); _tail_bn(t.bn_p[1], bns[3], zst[1], saves[3], train)
        _tail_bn(t.bn_c, bns[4], hst, saves[4], train)
        t.alpha_d[0], t.alpha_d[1], t.alpha_p[0], t.alpha_p[1] = ad1.data_ptr(), ad2.data_ptr(), ap1.data_ptr(), ap2.data_ptr()
        t.Wc, t.alpha_c, t.bres = wc.data_ptr(), ac.data_ptr(), bres.data_ptr()
        p = float(cfg.get("drop_p", 0.0)) if train else 0.0
        t.drop_p = p
        if p > 0.0:
            t.seed = seed_state(dev).data_ptr()
        t.salt[0], t.salt[1] = cfg["salts"]
        h0 = torch.empty(B, C, T, V, dtype=f32, device=dev)
        pooled = torch.empty(B, C, dtype=f32, device=dev)
        out = torch.empty(B, C, T, V, dtype=f32, device=dev)
        ostats = arena.take(2 * C * _lib.STAT_REPLICAS) if cfg.get("emit_stats") else None
        t.h0, t.pooled, t.out, t.ostats = h0.data_ptr(), pooled.data_ptr(), out.data_ptr(), _ptr(ostats)
        taps = cfg.get("taps")
        if taps is not None:                                           # branch records (diagnostics): five PReLU outputs
            for k in range(5):
                taps.append(torch.empty(B, C, T, V, dtype=f32, device=dev))
            t.tap_x[0], t.tap_x[1], t.tap_a[0], t.tap_a[1], t.tap_h = (v.data_ptr() for v in taps[-5:])
        stream = _stream(y1)
        for phase in (1, 2, 3):
            _lib.call("cg_dstd_tail_fwd", ctypes.byref(t), phase, stream)
        gate = torch.empty(B, C, dtype=f32, device=dev)
        H = sw1.shape[0]
        _lib.call("cg_se_gate_fwd", _ptr(pooled), _ptr(sw1), _ptr(sw2), _ptr(gate), B, C, H, stream)
        t.gate = gate.data_ptr()
        _lib.call("cg_dstd_tail_fwd", ctypes.byref(t), 4, stream)
        ctx.cfg, ctx.p = cfg, p
        ctx.seed_epoch = seed_epoch(dev) if p > 0.0 else None
        ctx.save_for_backward(*ts, h0, pooled, gate, saves)
        if ostats is not None:
            ctx.mark_non_differentiable(ostats)
        return out, ostats

    @staticmethod
    def backward(ctx, dout, _=None):
        sv = ctx.saved_tensors
        (y1, y2, r1, r2, w1, w2, gt1, bt1, ad1, gt2, bt2, ad2, gp1, bp1, ap1, gp2, bp2, ap2, wc, gc, bc, ac, sw1, sw2, bres) = sv[:25]
        h0, pooled, gate, saves = sv[25:]
        cfg = ctx.cfg
        _check_seed_epoch(ctx, y1.device, "dstd_tail")
        B, C, T, V = y1.shape
        dev, train, f32 = y1.device, bool(cfg["train"]), torch.float32
        dout = dout if dout.is_contiguous() else _copy(dout)
        t = _lib.DstdTail()
        t.B, t.C, t.T, t.V, t.train = B, C, T, V, 1 if train else 0
        t.y[0], t.y[1], t.r[0], t.r[1], t.w[0], t.w[1] = (v.data_ptr() for v in (y1, y2, r1, r2, w1, w2))
        bns = cfg["bn"]
        for dst, bn, k in ((t.bn_t[0], bns[0], 0), (t.bn_t[1], bns[1], 1), (t.bn_p[0], bns[2], 2), (t.bn_p[1], bns[3], 3), (t.bn_c, bns[4], 4)):
            _tail_bn(dst, bn, None, saves[k], False)
        t.alpha_d[0], t.alpha_d[1], t.alpha_p[0], t.alpha_p[1] = ad1.data_ptr(), ad2.data_ptr(), ap1.data_ptr(), ap2.data_ptr()
        t.Wc, t.alpha_c, t.bres, t.gate, t.h0 = wc.data_ptr(), ac.data_ptr(), bres.data_ptr(), gate.data_ptr(), h0.data_ptr()
        t.drop_p = ctx.p
        if ctx.p > 0.0:
            t.seed = seed_state(dev).data_ptr()
        t.salt[0], t.salt[1] = cfg["salts"]
        t.dout = dout.data_ptr()
        stream = _stream(y1)
        dgate = torch.empty(B, C, dtype=f32, device=dev)
        t.dgate = dgate.data_ptr()
        _lib.call("cg_dstd_tail_bwd", ctypes.byref(t), 1, stream)
        H = sw1.shape[0]
        dpooled = torch.empty(B, C, dtype=f32, device=dev)
        ws_floats = int(_lib.lib().cg_dstd_tail_ws_floats(C))
        zb, _ = _zeros(2 * C * H + ws_floats, dev)
        dsw1, dsw2 = zb[:C * H].view(sw1.shape), zb[C * H:2 * C * H].view(sw2.shape)
        _lib.call("cg_se_gate_bwd", _ptr(pooled), _ptr(sw1), _ptr(sw2), _ptr(gate), _ptr(dgate), _ptr(dpooled), _ptr(dsw1), _ptr(dsw2),
                  B, C, H, 1, stream)
        arena = _arena(dev)
        red_c = arena.take(2 * C + _lib.ALPHA_SLOTS)
        red_p = [arena.take(2 * C + _lib.ALPHA_SLOTS) for _ in range(2)]
        red_t = [arena.take(2 * C + _lib.ALPHA_SLOTS) for _ in range(2)]
        big = [torch.empty(B, C, T, V, dtype=f32, device=dev) for _ in range(6)]            # gp x2, dr x2, dy x2
        dws = [torch.empty(B, C, dtype=f32, device=dev) for _ in range(2)]
        small = torch.empty(15, C, dtype=f32, device=dev)   # dgamma/dbeta of tcn1,2 prelu1,2 compressor (10 rows) + 5 slope gradients
        dwc = torch.empty(C, 2 * C, dtype=f32, device=dev)
        t.dpooled, t.red_c = dpooled.data_ptr(), red_c.data_ptr()
        for i in range(2):
            t.gp[i], t.red_p[i], t.dr[i], t.dw[i], t.red_t[i], t.dy[i] = (big[i].data_ptr(), red_p[i].data_ptr(), big[2 + i].data_ptr(),
                                                                          dws[i].data_ptr(), red_t[i].data_ptr(), big[4 + i].data_ptr())
            t.dgamma_t[i], t.dbeta_t[i], t.dgamma_p[i], t.dbeta_p[i] = (small[2 * i].data_ptr(), small[2 * i + 1].data_ptr(),
                                                                        small[4 + 2 * i].data_ptr(), small[5 + 2 * i].data_ptr())
            t.dalpha_d[i], t.dalpha_p[i] = small[10 + i].data_ptr(), small[12 + i].data_ptr()
        t.dgamma_c, t.dbeta_c, t.dalpha_c = small[8].data_ptr(), small[9].data_ptr(), small[14].data_ptr()
        t.dWc_ws, t.dWc = zb[2 * C * H:].data_ptr(), dwc.data_ptr()
        for phase in (2, 3, 4, 5):
            _lib.call("cg_dstd_tail_bwd", ctypes.byref(t), phase, stream)
        al = lambda k: small[k, :1].reshape(1)
        grads = (big[4], big[5], big[2], big[3], dws[0], dws[1],
                 small[0], small[1], al(10), small[2], small[3], al(11),
                 small[4], small[5], al(12), small[6], small[7], al(13),
                 dwc.view(wc.shape), small[8], small[9], al(14), dsw1, dsw2, dout)
        return (None,) + tuple(g if ctx.needs_input_grad[1 + k] else None for k, g in enumerate(grads))


def dstd_tail(ys, ystats, rs, ws, bns, alphas, wc, se, bres, train, drop_p=0.0, salts=(0, 0), emit_stats=False, taps=None):
    """Fused tail of a DSTD_GC block.  ys / ystats: tcn outputs of the two Domain_GCNN layers and their f64 channel sums;
    rs: their residual addends; ws: gates (B,C); bns: (tcn1.1, tcn2.1, prelu1.0, prelu2.0, compressor.1) BatchNorm holders;
    alphas: (layer1.prelu, layer2.prelu, prelu1.1, prelu2.1, compressor.2) PReLU holders; wc: compressor conv weight
    (C,2C,1,1); se: SELayer2d holder; bres: block residual.  Returns (out, f64 channel sums of out or None)."""
    C = ys[0].shape[1]
    cfg = {"train": bool(train), "drop_p": float(drop_p), "salts": tuple(int(s) for s in salts), "ystats": tuple(ystats), "bn": tuple(bns),
           "emit_stats": bool(emit_stats), "taps": taps}
    ts = (ys[0], ys[1], rs[0], rs[1], ws[0], ws[1],
          bns[0].weight, bns[0].bias, alphas[0].weight, bns[1].weight, bns[1].bias, alphas[1].weight,
          bns[2].weight, bns[2].bias, alphas[2].weight, bns[3].weight, bns[3].bias, alphas[3].weight,
          wc.view(C, 2 * C), bns[4].weight, bns[4].bias, alphas[4].weight, se.w1, se.w2, bres)
    return _DstdTail.apply(cfg, *ts)


class _Map2AdjTail(torch.autograd.Function):
    """(s, q) of each tower pair -> Adj, CISTGCN.py:183-189 with the expansor of :165-170 (see csrc/map2adj_tail.hip).
    Tensor inputs per tower: s q W0 gamma beta alpha W4."""

    @staticmethod
    def _items(cfg, ts, train, saves, stats, e, dev, p):
        n = len(ts) // 7
        items = (_lib.AdjTail * n)()
        for i in range(n):
            s, q, w0, gamma, beta, alpha, w4 = ts[7 * i:7 * i + 7]
            B, V, T = s.shape
            t = items[i]
            dom = cfg["domains"][i]
            t.B, t.domain, t.train = B, dom, 1 if train else 0
            t.Kc, t.J = (V, T) if dom == 0 else (T, V)
            t.s, t.q, t.W0, t.alpha, t.W4 = s.data_ptr(), q.data_ptr(), w0.data_ptr(), alpha.data_ptr(), w4.data_ptr()
            _tail_bn(t.bn, cfg["bn"][i], stats[i] if stats is not None else None, saves[i], train and stats is not None)
            t.drop_p, t.salt = p, cfg["salts"][i]
            if p > 0.0:
                t.seed = seed_state(dev).data_ptr()
            t.e = e[i].data_ptr()
        return items

    @staticmethod
    def forward(ctx, cfg, *ts):
        ctx.set_materialize_grads(False)
        n = len(ts) // 7
        for i in range(n):
            for t in ts[7 * i:7 * i + 2]:
                _chk(t)
                if not t.is_contiguous():
                    raise ValueError("map2adj_tail expects contiguous tower outputs")
        dev, train, f32 = ts[0].device, bool(cfg["train"]), torch.float32
        p = float(cfg.get("drop_p", 0.0)) if train else 0.0
        arena = _arena(dev)
        shapes = []
        for i in range(n):
            B, V, T = ts[7 * i].shape
            Kc, J = (V, T) if cfg["domains"][i] == 0 else (T, V)
            shapes.append((B, Kc, J, J))
        saves = [torch.empty(2, sh[1], dtype=f32, device=dev) for sh in shapes]
        stats = [arena.take(2 * sh[1] * _lib.STAT_REPLICAS) for sh in shapes] if train else None
        e = [torch.empty(sh, dtype=f32, device=dev) for sh in shapes]
        adj = [torch.empty(sh, dtype=f32, device=dev) for sh in shapes]
        items = _Map2AdjTail._items(cfg, ts, train, saves, stats, e, dev, p)
        taps = cfg.get("taps")
        for i in range(n):
            items[i].adj = adj[i].data_ptr()
            if taps is not None:
                taps.append(torch.empty(shapes[i], dtype=f32, device=dev))
                items[i].tap = taps[-1].data_ptr()
        stream = _stream(ts[0])
        for phase in (1, 2):
            _lib.call("cg_map2adj_tail_fwd", items, n, phase, stream)
        ctx.cfg, ctx.p, ctx.n = cfg, p, n
        ctx.seed_epoch = seed_epoch(dev) if p > 0.0 else None
        ctx.save_for_backward(*ts, *e, *saves)
        return tuple(adj)

    @staticmethod
    def backward(ctx, *dadj):
        n, cfg = ctx.n, ctx.cfg
        sv = ctx.saved_tensors
        ts, e, saves = sv[:7 * n], sv[7 * n:8 * n], sv[8 * n:9 * n]
        dev, train, f32 = ts[0].device, bool(cfg["train"]), torch.float32
        if any(d is None for d in dadj):
            raise RuntimeError("map2adj_tail: every adjacency needs a gradient")
        _check_seed_epoch(ctx, dev, "map2adj_tail")
        dadj = [d if d.is_contiguous() else _copy(d) for d in dadj]
        items = _Map2AdjTail._items(cfg, ts, train, saves, None, e, dev, ctx.p)
        arena = _arena(dev)
        grads, keep = [], []
        stream = _stream(ts[0])
        for i in range(n):
            t = items[i]
            Kc = t.Kc
            ws = int(_lib.lib().cg_map2adj_tail_ws_floats(Kc))
            zb, _ = _zeros(ws, dev)
            red = arena.take(int(_lib.lib().cg_map2adj_tail_red_doubles(Kc)))
            g = torch.empty_like(e[i])
            part = torch.empty(int(_lib.lib().cg_map2adj_tail_part_floats(t.B, Kc, t.J)), dtype=f32, device=dev)
            t.part = part.data_ptr()
            keep += [g, zb, part]
            ds, dq = torch.empty_like(ts[7 * i]), torch.empty_like(ts[7 * i + 1])
            dw = torch.empty(2, Kc, Kc, dtype=f32, device=dev)
            small = torch.empty(3, Kc, dtype=f32, device=dev)
            t.dadj, t.g, t.red, t.ds, t.dq = dadj[i].data_ptr(), g.data_ptr(), red.data_ptr(), ds.data_ptr(), dq.data_ptr()
            t.dW0_ws, t.dW4_ws = zb[:ws // 2].data_ptr(), zb[ws // 2:].data_ptr()
            t.dW0, t.dW4, t.dgamma, t.dbeta, t.dalpha = dw[0].data_ptr(), dw[1].data_ptr(), small[0].data_ptr(), small[1].data_ptr(), small[2].data_ptr()
            grads += [ds, dq, dw[0].view(ts[7 * i + 2].shape), small[0], small[1], small[2, :1].reshape(1), dw[1].view(ts[7 * i + 6].shape)]
        for phase in (1, 2):
            _lib.call("cg_map2adj_tail_bwd", items, n, phase, stream)
        del keep
        return (None,) + tuple(g if ctx.needs_input_grad[1 + k] else None for k, g in enumerate(grads))


def map2adj_tail(seeds, expansors, train, drop_p=0.0, salts=(0, 0), taps=None):
    """Adjacency maps of a block's towers.  seeds: (domain 0|1, s (B,V,T), q (B,T,V)) per tower; expansors: their `expansor`
    holders (0: conv, 1: BatchNorm, 3: PReLU, 4: conv).  One launch per phase for all towers."""
    cfg = {"train": bool(train), "drop_p": float(drop_p), "salts": tuple(int(s) for s in salts), "domains": tuple(int(d) for d, _, _ in seeds),
           "bn": tuple(e[1] for e in expansors), "taps": taps}
    ts = []
    for (dom, s, q), e in zip(seeds, expansors):
        kc = e[0].out_channels
        ts += [s, q, e[0].weight.view(kc, kc), e[1].weight, e[1].bias, e[3].weight, e[4].weight.view(kc, kc)]
    return _Map2AdjTail.apply(cfg, *ts)


def block_input_ok(x):
    """True when `block_input` takes the block input x (B,C,T,V)."""
    if x.dim() != 4 or not x.is_contiguous() or x.dtype != torch.float32:
        return False
    B, C, T, V = x.shape
    return bool(_lib.lib().cg_block_input_supported(B, C, T, V))


_BLOCK_INPUT_MAXG = 8


class _BlockInput(torch.autograd.Function):
    """xn = global_norm(x) and _get_stats_(xn) of a DSTD_GC block (CISTGCN.py:375-379, :360-371; csrc/block_input.hip).  Outputs: `n`
    aliases of xn (one per consumer, as ops.fanout) and two of the statistics; backward takes the consumers' gradients as they are:
    fan-in sum, statistics backward and BatchNorm backward are two streaming passes.  Tensor inputs: x gamma beta."""

    @staticmethod
    def _block(x, bn, save, train):
        B, C, T, V = x.shape
        t = _lib.BlockInput()
        t.B, t.C, t.T, t.V, t.train = B, C, T, V, 1 if train else 0
        t.x = x.data_ptr()
        _tail_bn(t.bn, bn, None, save, train)
        return t

    @staticmethod
    def forward(ctx, cfg, x, gamma, beta):
        ctx.set_materialize_grads(False)
        _chk(x)
        dev, f32, train, n = x.device, torch.float32, bool(cfg["train"]), int(cfg["n"])
        B, C, T, V = x.shape
        save = torch.empty(2, C, dtype=f32, device=dev)
        t = _BlockInput._block(x, cfg["bn"], save, train)
        stream = _stream(x)
        if train:
            if B * T * V <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
            stats = cfg.get("stats")
            if stats is None:
                stats = _arena(dev).take(2 * C * _lib.STAT_REPLICAS)
                it = _lib.StatsArgs()
                it.x, it.xv, it.pre, it.stats = x.data_ptr(), _view4(x), None, stats.data_ptr()
                _lib.call("cg_chan_stats_many", (_lib.StatsArgs * 1)(it), 1, stream)
            t.bn.stats = stats.data_ptr()
        xn = torch.empty_like(x)
        rows = torch.empty(2, B, C, T, dtype=f32, device=dev)
        out = torch.empty(B, 2 + 2 * T, dtype=f32, device=dev)
        t.xn, t.rm, t.rq, t.out = xn.data_ptr(), rows[0].data_ptr(), rows[1].data_ptr(), out.data_ptr()
        _lib.call("cg_block_input_fwd", ctypes.byref(t), stream)
        ctx.cfg = cfg
        ctx.save_for_backward(x, gamma, beta, rows, save)
        return tuple(xn.view_as(xn) for _ in range(n)) + (out.view_as(out), out.view_as(out))

    @staticmethod
    def backward(ctx, *grads):
        x, gamma, beta, rows, save = ctx.saved_tensors
        cfg = ctx.cfg
        n = int(cfg["n"])
        gs = [g for g in grads[:n] if g is not None]
        douts = [g for g in grads[n:n + 2] if g is not None]
        if not gs and not douts:
            return None, None, None, None
        dev, f32 = x.device, torch.float32
        B, C, T, V = x.shape
        gs = [g if g.is_contiguous() else _copy(g) for g in gs]
        douts = [g if g.stride(1) == 1 else _copy(g) for g in douts]       # column slices of the gate inputs' gradient are taken as they are
        while len(gs) > _BLOCK_INPUT_MAXG:                      # more consumers than pointer slots: fold the tail first
            gs = gs[:_BLOCK_INPUT_MAXG - 1] + [_sum_tensors(gs[_BLOCK_INPUT_MAXG - 1:])]
        t = _BlockInput._block(x, cfg["bn"], save, False)
        t.train = 1 if cfg["train"] else 0
        t.rm, t.rq = rows[0].data_ptr(), rows[1].data_ptr()
        t.ng = len(gs)
        for i, g in enumerate(gs):
            t.g[i] = g.data_ptr()
        for i, g in enumerate(douts):
            t.dout[i], t.dout_ld[i] = g.data_ptr(), g.stride(0)
        pq = torch.empty(B, C, T, 2, dtype=f32, device=dev) if douts else None
        gsum = torch.empty_like(x)
        dx = torch.empty_like(x)
        small = torch.empty(2, C, dtype=f32, device=dev)
        red = _arena(dev).take(2 * C * _lib.STAT_REPLICAS)
        t.pq, t.gsum, t.red, t.dx, t.dgamma, t.dbeta = _ptr(pq), gsum.data_ptr(), red.data_ptr(), dx.data_ptr(), small[0].data_ptr(), small[1].data_ptr()
        _lib.call("cg_block_input_bwd", ctypes.byref(t), _stream(x))
        del gs, douts, gsum, pq
        return (None, dx if ctx.needs_input_grad[1] else None, small[0] if ctx.needs_input_grad[2] else None,
                small[1] if ctx.needs_input_grad[3] else None)


def block_input(x, bn, train, n, stats=None):
    """(aliases of xn = bn(x) [n of them, one per consumer], (stats_a, stats_b) = two aliases of _get_stats_(xn)) for the block input x
    (B,C,T,V); `stats`: f64 channel sums of x when the producer emitted them (train mode)."""
    cfg = {"train": bool(train), "n": int(n), "bn": bn, "stats": stats}
    out = _BlockInput.apply(cfg, x, bn.weight, bn.bias)
    return list(out[:n]), (out[n], out[n + 1])


def gate_head_ok(B, C, S, prelus):
    return bool(_lib.lib().cg_gate_head_supported(B, C, S)) and all(p.weight.numel() == 1 for p in prelus)


class _GateHead(torch.autograd.Function):
    """Tail of the gate paths of a DSTD_GC block, CISTGCN.py:337-352 / :378-384 (csrc/gate_head.hip): per path
    z (B,C) -> BatchNorm -> Dropout -> PReLU -> cat(block statistics) -> Linear -> BatchNorm1d -> Dropout -> PReLU -> Linear = w (B,C).
    Tensor inputs: per path z, stats, gamma2, beta2, alpha2, Wl, gamma3, beta3, alpha3, W2 (10 each)."""

    @staticmethod
    def _block(cfg, ts, n, train, saves, ys, dev):
        B, C = ts[0].shape
        S = ts[1].shape[1]
        t = _lib.GateHead()
        t.B, t.C, t.S, t.train, t.n = B, C, S, 1 if train else 0, n
        p = float(cfg.get("drop_p", 0.0)) if train else 0.0
        t.drop_p = p
        if p > 0.0:
            t.seed = seed_state(dev).data_ptr()
        for i in range(n):
            z, stats, g2, b2, a2, wl, g3, b3, a3, w2 = ts[10 * i:10 * i + 10]
            q = t.p[i]
            q.z, q.stats, q.stats_ld = z.data_ptr(), stats.data_ptr(), stats.stride(0)
            _tail_bn(q.bn2, cfg["bn2"][i], None, saves[i][0], train)
            _tail_bn(q.bn3, cfg["bn3"][i], None, saves[i][1], train)
            q.alpha2, q.Wl, q.alpha3, q.W2 = a2.data_ptr(), wl.data_ptr(), a3.data_ptr(), w2.data_ptr()
            q.salt2, q.salt3 = cfg["salts"][i]
            q.y = ys[i].data_ptr()
        return t, p

    @staticmethod
    def forward(ctx, cfg, n, *ts):
        ctx.set_materialize_grads(False)
        dev, f32, train = ts[0].device, torch.float32, bool(cfg["train"])
        B, C = ts[0].shape
        for i in range(n):
            z, stats = ts[10 * i], ts[10 * i + 1]
            _chk(z), _chk(stats)
            if not z.is_contiguous() or stats.stride(1) != 1 or not ts[10 * i + 5].is_contiguous() or not ts[10 * i + 9].is_contiguous():
                raise ValueError("gate_head expects contiguous z / weights and unit-stride statistics rows")
            if train and B < 2:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(z.shape),))
        saves = [torch.empty(2, 2, C, dtype=f32, device=dev) for _ in range(n)]
        ys = [torch.empty(B, C, dtype=f32, device=dev) for _ in range(n)]
        ws = [torch.empty(B, C, dtype=f32, device=dev) for _ in range(n)]
        t, p = _GateHead._block(cfg, ts, n, train, saves, ys, dev)
        taps = cfg.get("taps")
        for i in range(n):
            t.p[i].w = ws[i].data_ptr()
            if taps is not None:
                taps += [torch.empty(B, C, dtype=f32, device=dev), torch.empty(B, C, dtype=f32, device=dev)]
                t.p[i].tap2, t.p[i].tap3 = taps[-2].data_ptr(), taps[-1].data_ptr()
        _lib.call("cg_gate_head_fwd", ctypes.byref(t), _stream(ts[0]))
        ctx.cfg, ctx.n, ctx.p = cfg, n, p
        ctx.seed_epoch = seed_epoch(dev) if p > 0.0 else None
        ctx.save_for_backward(*ts, *ys, *saves)
        return tuple(ws)

    @staticmethod
    def backward(ctx, *dws):
        n, cfg = ctx.n, ctx.cfg
        sv = ctx.saved_tensors
        ts, ys, saves = sv[:10 * n], sv[10 * n:11 * n], sv[11 * n:12 * n]
        dev, f32, train = ts[0].device, torch.float32, bool(cfg["train"])
        _check_seed_epoch(ctx, dev, "gate_head")
        B, C = ts[0].shape
        S = ts[1].shape[1]
        t, _ = _GateHead._block(cfg, ts, n, train, saves, ys, dev)
        keep, grads = [], []
        nscr = int(_lib.lib().cg_gate_head_scratch_floats(B, C, S))
        for i in range(n):
            d = dws[i]
            if d is None:
                d = _zeros(B * C, dev)[0].view(B, C)
            d = d if d.is_contiguous() else _copy(d)
            dz = torch.empty(B, C, dtype=f32, device=dev)
            dst = torch.empty(B, S, dtype=f32, device=dev)
            zw, _ = _zeros(C * (C + S) + C * C, dev)                         # accumulated with float atomics over the chunks of the batch
            dwl, dw2 = zw[:C * (C + S)].view(C, C + S), zw[C * (C + S):].view(C, C)
            small = torch.empty(6, C, dtype=f32, device=dev)
            scr = torch.empty(max(nscr, 1), dtype=f32, device=dev)
            q = t.p[i]
            q.red = _arena(dev).take(2).data_ptr()
            q.dw, q.dz, q.dstats, q.dWl, q.dW2, q.scratch = d.data_ptr(), dz.data_ptr(), dst.data_ptr(), dwl.data_ptr(), dw2.data_ptr(), scr.data_ptr()
            q.dgamma2, q.dbeta2, q.dalpha2, q.dgamma3, q.dbeta3, q.dalpha3 = (small[k].data_ptr() for k in range(6))
            keep += [d, scr]
            grads += [dz.view(ts[10 * i].shape), dst, small[0], small[1], small[2, :1].reshape(1), dwl.view(ts[10 * i + 5].shape),
                      small[3], small[4], small[5, :1].reshape(1), dw2.view(ts[10 * i + 9].shape)]
        _lib.call("cg_gate_head_bwd", ctypes.byref(t), _stream(ts[0]))
        del keep
        return (None, None) + tuple(g if ctx.needs_input_grad[2 + k] else None for k, g in enumerate(grads))


def gate_head(zs, stats, convs, maps, train, drop_p=0.0, salts=((0, 0), (0, 0)), taps=None):
    """Gates w_i (B,C) of a block's gate paths from z_i (B,C) = output of conv_i[4] and stats_i (B,S): `convs[i]` / `maps[i]` are the
    reference's conv_s|t (slots 5: BatchNorm2d, 7: PReLU used) and map_s|t (0: Linear, 1: BatchNorm1d, 3: PReLU, 4: Linear) holders;
    salts[i] = the site ids of the two Dropout layers of path i.  One launch for all paths, forward and backward."""
    n = len(zs)
    cfg = {"train": bool(train), "drop_p": float(drop_p), "salts": tuple((int(a), int(b)) for a, b in salts),
           "bn2": tuple(c[5] for c in convs), "bn3": tuple(m[1] for m in maps), "taps": taps}
    ts = []
    for z, st, c, m in zip(zs, stats, convs, maps):
        ts += [z, st, c[5].weight, c[5].bias, c[7].weight, m[0].weight, m[1].weight, m[1].bias, m[3].weight, m[4].weight]
    return list(_GateHead.apply(cfg, n, *ts))


def context_heads_ok(x, hidden):
    """True when `context_heads` takes the two one-channel heads of x (B,1,H,W)."""
    return x.dim() == 4 and x.shape[1] == 1 and x.is_contiguous() and hidden <= 64 and x.shape[2] * x.shape[3] <= 16384


class _ContextHeads(torch.autograd.Function):
    """(max_p, mean_p) of PReLU(BN(w_h x)) for the two 1 -> C heads of the ContextLayer, CISTGCN.py:408-418 / :465 / :467
    (see csrc/context_heads.hip).  Tensor inputs: x | w gamma beta alpha of head 0 | of head 1."""

    @staticmethod
    def _block(x, ts, bns, saves, train):
        B, _, H, W = x.shape
        t = _lib.CtxHeads()
        t.B, t.P, t.C, t.train = B, H * W, ts[0].numel(), 1 if train else 0
        t.x = x.data_ptr()
        for h in range(2):
            w, gamma, beta, alpha = ts[4 * h:4 * h + 4]
            t.w[h], t.alpha[h] = w.data_ptr(), alpha.data_ptr()
            _tail_bn(t.bn[h], bns[h], None, saves[h], train)
        return t

    @staticmethod
    def forward(ctx, cfg, x, *ts):
        ctx.set_materialize_grads(False)
        _chk(x)
        dev, f32, train = x.device, torch.float32, bool(cfg["train"])
        B, _, H, W = x.shape
        C = ts[0].numel()
        saves = torch.empty(2, 2, C, dtype=f32, device=dev)
        xsave = torch.empty(2, dtype=f32, device=dev)
        t = _ContextHeads._block(x, ts, cfg["bn"], saves, train)
        stream = _stream(x)
        if train:
            if B * H * W <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
            xstats = _arena(dev).take(2 * _lib.STAT_REPLICAS)
            it = _lib.StatsArgs()
            it.x, it.xv, it.pre, it.stats = x.data_ptr(), _view4(x), None, xstats.data_ptr()
            _lib.call("cg_chan_stats_many", (_lib.StatsArgs * 1)(it), 1, stream)
            t.xstats = xstats.data_ptr()
        y = torch.empty(2, B, C, dtype=f32, device=dev)
        arg = torch.empty(B, C, dtype=torch.int32, device=dev)
        t.y[0], t.y[1], t.arg, t.xsave = y[0].data_ptr(), y[1].data_ptr(), arg.data_ptr(), xsave.data_ptr()
        taps = cfg.get("taps")
        if taps is not None:
            for h in range(2):
                taps.append(torch.empty(B, C, H, W, dtype=f32, device=dev))
                t.tap[h] = taps[-1].data_ptr()
        _lib.call("cg_context_heads_fwd", ctypes.byref(t), stream)
        ctx.cfg = cfg
        ctx.save_for_backward(x, *ts, arg, saves, xsave)
        ctx.mark_non_differentiable(arg)
        return y[0], y[1]

    @staticmethod
    def backward(ctx, dy0, dy1):
        sv = ctx.saved_tensors
        x, ts, arg, saves, xsave = sv[0], sv[1:9], sv[9], sv[10], sv[11]
        dev, f32, train = x.device, torch.float32, bool(ctx.cfg["train"])
        B, _, H, W = x.shape
        C = ts[0].numel()
        t = _ContextHeads._block(x, ts, ctx.cfg["bn"], saves, False)
        t.train = 1 if train else 0
        t.arg, t.xsave = arg.data_ptr(), xsave.data_ptr()
        dys = []
        for h, d in enumerate((dy0, dy1)):
            if d is None:                                         # a head that does not reach the loss
                d = _zeros(B * C, dev)[0].view(B, C)
            d = d if d.is_contiguous() else _copy(d)
            dys.append(d)
            t.dy[h] = d.data_ptr()
        red = _arena(dev).take(int(_lib.lib().cg_context_heads_red_doubles(C)))
        dx = torch.empty_like(x)
        small = torch.empty(2, 4, C, dtype=f32, device=dev)     # per head: dw, dgamma, dbeta, dalpha (one number)
        t.red, t.dx = red.data_ptr(), dx.data_ptr()
        for h in range(2):
            t.dw[h], t.dgamma[h], t.dbeta[h], t.dalpha[h] = (small[h, k].data_ptr() for k in range(4))
        _lib.call("cg_context_heads_bwd", ctypes.byref(t), _stream(x))
        del dys
        grads = []
        for h in range(2):
            grads += [small[h, 0].view(ts[4 * h].shape), small[h, 1], small[h, 2], small[h, 3, :1].reshape(1)]
        return (None, dx if ctx.needs_input_grad[1] else None) + tuple(g if ctx.needs_input_grad[2 + k] else None for k, g in enumerate(grads))


def context_heads(x, head_max, head_mean, train, taps=None):
    """(max over positions, mean over positions) of the two Conv2d(1, C, 1) -> BatchNorm2d -> PReLU heads `head_max` / `head_mean`
    (holders with [0] conv, [1] BatchNorm, [2] PReLU) of x (B,1,H,W); the (B,C,H,W) activations are never stored."""
    cfg = {"train": bool(train), "bn": (head_max[1], head_mean[1]), "taps": taps}
    ts = []
    for hd in (head_max, head_mean):
        ts += [hd[0].weight, hd[1].weight, hd[1].bias, hd[2].weight]
    return _ContextHeads.apply(cfg, x, *ts)


def pointwise_maps_ok(x, weights):
    """True when `pointwise_maps` takes this problem (else the caller batches the maps through the generic contraction)."""
    if x.dim() != 4 or not x.is_contiguous() or len(weights) > 4:
        return False
    B, C, H, W = x.shape
    rows = sum((w.shape[0] + 15) // 16 * 16 for w in weights)
    return C <= 128 and (H * W) % 2 == 0 and rows <= 128 and (rows // 16) * ((C + 15) // 16) <= 32 and all(w.shape[0] <= 64 for w in weights)


class _PointwiseMaps(torch.autograd.Function):
    """y_i = W_i x (+ bias_i) for up to four 1x1 convolutions of one input, CISTGCN.py:138-163 / :183-186 and the residual maps
    :246-254 / :357-365 (see csrc/tower_maps.hip).  Tensor inputs: x, W_1 .. W_n, bias_1 .. bias_n (None where a map has none);
    outputs y_1 .. y_n, then their f64 channel sums (or None)."""

    @staticmethod
    def _block(x, ws, bs):
        B, C, H, W = x.shape
        t = _lib.PwMaps()
        t.B, t.Cin, t.P, t.n = B, C, H * W, len(ws)
        t.x = x.data_ptr()
        for i, w in enumerate(ws):
            t.W[i], t.M[i] = w.data_ptr(), w.shape[0]
            t.bias[i] = _ptr(bs[i])
        return t

    @staticmethod
    def forward(ctx, want_stats, n, x, *wb):
        ctx.set_materialize_grads(False)
        _chk(x)
        B, C, H, W = x.shape
        dev, f32 = x.device, torch.float32
        ws = [w if w.is_contiguous() else _copy(w) for w in wb[:n]]
        bs = [None if b is None else (b if b.is_contiguous() else _copy(b)) for b in wb[n:]]
        t = _PointwiseMaps._block(x, ws, bs)
        ys = [torch.empty(B, w.shape[0], H, W, dtype=f32, device=dev) for w in ws]
        stats = [_arena(dev).take(2 * w.shape[0] * _lib.STAT_REPLICAS) for w in ws] if want_stats else [None] * len(ws)
        for i in range(len(ws)):
            t.y[i], t.stats[i] = ys[i].data_ptr(), _ptr(stats[i])
        _lib.call("cg_pointwise_maps_fwd", ctypes.byref(t), _stream(x))
        ctx.save_for_backward(x, *ws)
        ctx.has_bias = [b is not None for b in bs]
        if want_stats:
            ctx.mark_non_differentiable(*stats)
        return tuple(ys) + tuple(stats)

    @staticmethod
    def backward(ctx, *grads):
        x, *ws = ctx.saved_tensors
        n = len(ws)
        dys = grads[:n]
        if any(d is None for d in dys):
            raise RuntimeError("pointwise_maps: every map needs a gradient")
        dys = [d if d.is_contiguous() else _copy(d) for d in dys]
        dev, f32 = x.device, torch.float32
        t = _PointwiseMaps._block(x, ws, [None] * n)
        dx = torch.empty_like(x)
        dws = [torch.empty_like(w) for w in ws]
        dbs = [torch.empty(w.shape[0], dtype=f32, device=dev) if hb else None for w, hb in zip(ws, ctx.has_bias)]
        zb, _ = _zeros(int(_lib.lib().cg_pointwise_maps_ws_floats(x.shape[1])), dev)
        for i in range(n):
            t.dy[i], t.dW[i], t.db[i] = dys[i].data_ptr(), dws[i].data_ptr(), _ptr(dbs[i])
        t.dx, t.dW_ws = dx.data_ptr(), zb.data_ptr()
        _lib.call("cg_pointwise_maps_bwd", ctypes.byref(t), _stream(x))
        del dys
        return (None, None, dx if ctx.needs_input_grad[2] else None) + tuple(dw if ctx.needs_input_grad[3 + i] else None for i, dw in enumerate(dws)) \
            + tuple(db if (db is not None and ctx.needs_input_grad[3 + n + i]) else None for i, db in enumerate(dbs))


def pointwise_maps(x, weights, want_stats=False, biases=None):
    """[(y_i, channel sums or None)] of the 1x1 maps `weights` (each (M_i, C), optional `biases` (M_i) or None) of x (B,C,H,W): one read
    of x forward, one read of x and of every gradient backward.  Shapes outside `pointwise_maps_ok` are the caller's business."""
    n = len(weights)
    biases = [None] * n if biases is None else list(biases)
    out = _PointwiseMaps.apply(bool(want_stats), n, x, *weights, *biases)
    return [(out[i], out[n + i] if want_stats else None) for i in range(n)]


_ONES = {}


def _one(dev):
    """a constant slope 1 (PReLU = identity) for the maps of `tower_maps` that have no PReLU behind their BatchNorm"""
    if dev not in _ONES:
        _ONES[dev] = torch.ones(1, dtype=torch.float32, device=dev)
    return _ONES[dev]


class _TowerMaps(torch.autograd.Function):
    """h_i = PReLU(BN(W_i x + b_i)) for up to four 1x1 maps of one input followed by BatchNorm2d (+ PReLU): the first level of the
    Map2Adj towers of a block (CISTGCN.py:138-141 / :156-158 applied by :183-186; no bias, PReLU) and the residual maps of a block that
    changes its width (:246-254 / :357-365: bias, no PReLU).  Forward = cg_pointwise_maps_fwd + cg_norm_act_fwd_many, as
    the two operators do; backward: ONE reduction pass over (dh_i, y_i) and the pointwise backward undoing BatchNorm and PReLU while
    it loads dh_i - the gradient in front of the BatchNorm is never stored (as two operators: reduce + apply passes of cg_norm_act_bwd,
    432 MB written and read again per block at B = 256).  Tensor inputs: x | W_1..n | gamma_1..n | beta_1..n | alpha_1..n | b_1..n
    (alpha_i / b_i None: no PReLU / no bias)."""

    @staticmethod
    def forward(ctx, cfg, n, x, *ts):
        ctx.set_materialize_grads(False)
        _chk(x)
        B, C, H, W = x.shape
        dev, f32, train = x.device, torch.float32, bool(cfg["train"])
        ws = [w if w.is_contiguous() else _copy(w) for w in ts[:n]]
        gammas, betas, alphas = ts[n:2 * n], ts[2 * n:3 * n], ts[3 * n:4 * n]
        bs = [None if b is None else (b if b.is_contiguous() else _copy(b)) for b in ts[4 * n:5 * n]]
        t = _PointwiseMaps._block(x, ws, bs)
        ys = [torch.empty(B, w.shape[0], H, W, dtype=f32, device=dev) for w in ws]
        stats = [_arena(dev).take(2 * w.shape[0] * _lib.STAT_REPLICAS) for w in ws] if train else [None] * n
        for i in range(n):
            t.y[i], t.stats[i] = ys[i].data_ptr(), _ptr(stats[i])
        stream = _stream(x)
        _lib.call("cg_pointwise_maps_fwd", ctypes.byref(t), stream)
        if cfg.get("defer") is not None:
            # the BatchNorm / PReLU of every map is applied by its consumer while it loads (collapse_rows / collapse_cols with `transform`):
            # the outputs handed on are the RAW maps, the consumers fill `save` (mean, rstd) and update the running statistics, and their
            # backward returns the gradient with respect to the ACTIVATED tensor - exactly what this function's backward expects
            hs = ys
            saves = [torch.empty(2, w.shape[0], dtype=f32, device=dev) for w in ws]
            for i in range(n):
                if alphas[i] is None or alphas[i].numel() != 1:
                    raise ValueError("tower_maps(defer=True) needs a shared-slope PReLU behind every map")
                cfg["defer"].append({"bn": cfg["bn"][i], "gamma": gammas[i], "beta": betas[i], "alpha": alphas[i], "stats": stats[i],
                                     "save": saves[i], "train": train, "consumed": False})
            ctx.deferred = cfg["defer"]
        else:
            arr = (NormAct * n)()
            hs, saves, pending = [], [], []
            for i in range(n):
                h, save, _, _ = _na_fill_fwd(arr[i], ys[i], None, None, gammas[i], betas[i], alphas[i],
                                             {"bn": cfg["bn"][i], "train": train, "stats": stats[i]}, pending)
                hs.append(h); saves.append(save)
            assert not pending
            _lib.call("cg_norm_act_fwd_many", arr, n, stream)
        ctx.cfg, ctx.n = {"train": cfg["train"], "bn": cfg["bn"]}, n
        ctx.has_alpha, ctx.has_bias = [a is not None for a in alphas], [b is not None for b in bs]
        ctx.save_for_backward(x, *ws, *gammas, *betas, *[a if a is not None else _one(dev) for a in alphas], *ys, *saves)
        return tuple(hs)

    @staticmethod
    def backward(ctx, *dhs):
        n, cfg = ctx.n, ctx.cfg
        sv = ctx.saved_tensors
        x, ws, gammas, betas, alphas, ys, saves = sv[0], sv[1:1 + n], sv[1 + n:1 + 2 * n], sv[1 + 2 * n:1 + 3 * n], sv[1 + 3 * n:1 + 4 * n], sv[1 + 4 * n:1 + 5 * n], sv[1 + 5 * n:1 + 6 * n]
        if any(d is None for d in dhs):
            raise RuntimeError("tower_maps: every map needs a gradient")
        if any(not tr["consumed"] for tr in getattr(ctx, "deferred", ())):
            raise RuntimeError("tower_maps(defer=True): a raw map was not consumed by collapse_rows / collapse_cols with its transform "
                               "(its BatchNorm statistics were never finalised)")
        dhs = [d if d.is_contiguous() else _copy(d) for d in dhs]
        dev, f32, train = x.device, torch.float32, bool(cfg["train"])
        stream = _stream(x)
        # pass 1: channel sums of the gradient in front of the BatchNorm, slope gradients, dgamma / dbeta
        arr = (NormAct * n)()
        reds, smalls, have_red = [], [], []
        for i in range(n):
            a, bn = arr[i], cfg["bn"][i]
            M = ws[i].shape[0]
            a.x, a.xv, a.dy, a.dyv = ys[i].data_ptr(), _view4(ys[i]), dhs[i].data_ptr(), _view4(dhs[i])
            a.bn_mode = 1 if train else 2
            a.gamma, a.beta = gammas[i].data_ptr(), betas[i].data_ptr()
            a.running_mean, a.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            a.momentum, a.eps = bn.momentum, bn.eps
            a.save_mean, a.save_rstd = saves[i][0].data_ptr(), saves[i][1].data_ptr()
            nal = alphas[i].numel() if ctx.has_alpha[i] else 0
            if nal:
                a.alpha, a.alpha_n = alphas[i].data_ptr(), nal
            given = ctx.deferred[i].get("red") if getattr(ctx, "deferred", None) else None
            red = given if given is not None else _arena(dev).take(2 * M + (_lib.ALPHA_SLOTS if nal <= 1 else nal))
            have_red.append(given is not None)
            small = torch.empty(3, M, dtype=f32, device=dev)
            a.red, a.dgamma, a.dbeta = red.data_ptr(), small[0].data_ptr(), small[1].data_ptr()
            if nal:
                a.dalpha = small[2].data_ptr()
            reds.append(red); smalls.append(small)
        if all(have_red):
            _lib.call("cg_norm_act_params_many", arr, n, stream)        # the consumers' backward kernels summed already: parameter gradients only
        else:
            if any(have_red):                                           # mixed: redo all of them here (fresh words for the ones already summed)
                for i in range(n):
                    if have_red[i]:
                        nal = alphas[i].numel() if ctx.has_alpha[i] else 0
                        reds[i] = _arena(dev).take(2 * ws[i].shape[0] + (_lib.ALPHA_SLOTS if nal <= 1 else nal))
                        arr[i].red = reds[i].data_ptr()
            _lib.call("cg_norm_act_bwd_reduce_many", arr, n, stream)
        # pass 2: dx and dW_i (db_i) with BatchNorm / PReLU undone on load
        t = _PointwiseMaps._block(x, ws, [None] * n)
        dx = torch.empty_like(x)
        dws = [torch.empty_like(w) for w in ws]
        dbs = [torch.empty(w.shape[0], dtype=f32, device=dev) if hb else None for w, hb in zip(ws, ctx.has_bias)]
        zb, _ = _zeros(int(_lib.lib().cg_pointwise_maps_ws_floats(x.shape[1])), dev)
        for i in range(n):
            t.dy[i], t.dW[i], t.db[i] = dhs[i].data_ptr(), dws[i].data_ptr(), _ptr(dbs[i])
            t.yraw[i], t.bn_save[i], t.bn_gamma[i], t.bn_beta[i] = ys[i].data_ptr(), saves[i].data_ptr(), gammas[i].data_ptr(), betas[i].data_ptr()
            t.bn_red[i], t.prelu[i] = reds[i].data_ptr(), alphas[i].data_ptr()           # slope 1 where there is no PReLU
        t.bn_train = 1 if train else 0
        t.dx, t.dW_ws = dx.data_ptr(), zb.data_ptr()
        _lib.call("cg_pointwise_maps_bwd", ctypes.byref(t), stream)
        del dhs
        need = ctx.needs_input_grad
        grads = [dx if need[2] else None] + [dws[i] if need[3 + i] else None for i in range(n)]
        grads += [smalls[i][0] if need[3 + n + i] else None for i in range(n)]
        grads += [smalls[i][1] if need[3 + 2 * n + i] else None for i in range(n)]
        grads += [smalls[i][2, :1].reshape(alphas[i].shape) if (ctx.has_alpha[i] and need[3 + 3 * n + i]) else None for i in range(n)]
        grads += [dbs[i] if (dbs[i] is not None and need[3 + 4 * n + i]) else None for i in range(n)]
        return (None, None) + tuple(grads)


def tower_maps_ok(x, weights, prelus):
    return pointwise_maps_ok(x, weights) and all(p is None or p.weight.numel() == 1 for p in prelus)


def tower_maps(x, weights, bns, prelus, train, biases=None, defer=False):
    """[h_i] = PReLU(BN(W_i x + b_i)) for the 1x1 maps `weights` (each (M_i, C)) of x (B,C,H,W) with their BatchNorm2d / PReLU holders
    (`prelus[i]` None: no PReLU; `biases[i]` None: no bias).
    defer=True (every map with a shared-slope PReLU): returns ([raw maps y_i], [transform_i]) instead - EVERY y_i must then be consumed by
    exactly one `collapse_rows(y_i, w, stats, transform=transform_i)` or `collapse_cols(...)`, which applies BatchNorm + PReLU while it
    loads (the activated tensors are never stored)."""
    n = len(weights)
    biases = [None] * n if biases is None else list(biases)
    cfg = {"train": bool(train), "bn": tuple(bns)}
    if defer:
        cfg["defer"] = []
    out = list(_TowerMaps.apply(cfg, n, x, *weights, *[b.weight for b in bns], *[b.bias for b in bns],
                                *[None if p is None else p.weight for p in prelus], *biases))
    return (out, cfg["defer"]) if defer else out


class _SplitChannels(torch.autograd.Function):
    """Inverse of `cat_channels`: channel ranges of one tensor as views (no copy); backward gathers the pieces' gradients."""

    @staticmethod
    def forward(ctx, sizes, y):
        ctx.set_materialize_grads(False)
        ctx.sizes, ctx.shape = sizes, tuple(y.shape)
        outs, c0 = [], 0
        for c in sizes:
            outs.append(y.narrow(1, c0, c))
            c0 += c
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        dev = next(g for g in gs if g is not None).device
        dy = torch.empty(ctx.shape, dtype=torch.float32, device=dev)
        c0, items = 0, []
        for c, g in zip(ctx.sizes, gs):
            dst = dy.narrow(1, c0, c)
            c0 += c
            if g is None:
                raise RuntimeError("split_channels: every piece needs a gradient")
            it = _lib.CopyItem()
            it.y, it.yv, it.a, it.av = dst.data_ptr(), _view4(dst), g.data_ptr(), _view4(g)
            items.append(it)
        for i0 in range(0, len(items), 4):
            chunk = items[i0:i0 + 4]
            _lib.call("cg_copy_many", (_lib.CopyItem * len(chunk))(*chunk), len(chunk), _stream(dy))
        return None, dy


def split_channels(y, sizes):
    """Views of consecutive channel ranges of y (B,C,...), differentiable (the gradients are gathered by one copy launch)."""
    return _SplitChannels.apply(tuple(int(s) for s in sizes), y)


_ROWS_KERNELS = __import__("os").environ.get("CISTGCN_ROWS_KERNELS", "1") != "0"     # 0: the generic contraction (tuning aid)


# BatchNorm / PReLU backward sums of a deferred tower map from the collapsing backward kernel (`in_red`) instead of cg_norm_act_bwd_reduce_many.
# OFF by default: measured at the headline shape the two backward kernels grow from 48 / 55 to 71 / 82 us per call (228 / 286 VGPRs: the raw
# x of the result positions travels a sample ahead, nine f64 sums per lane), more than the 65 us reduction pass per block they replace.
# `transform["fold_reduce"] = True` selects it per map (tests/checks.py::check_tower_collapse does for one shape).
_FOLD_REDUCE = os.environ.get("CISTGCN_FOLD_REDUCE", "0") == "1"


def collapse_rows_ok(x, w):
    """True when `collapse_rows` takes the (T,1) convolution `w` (O,C,T) of x (B,C,T,V)."""
    if not _ROWS_KERNELS or x.dim() != 4 or not x.is_contiguous() or w.dim() != 3:
        return False
    B, C, T, V = x.shape
    return w.shape[1] == C and w.shape[2] == T and V <= 32 and w.shape[0] <= 64 and (C * T) % 4 == 0


class _CollapseRows(torch.autograd.Function):
    """y[b,o,v] = sum_{c,t} w[o,c,t] x[b,c,t,v] (nn.Conv2d(C, O, (T,1)), CISTGCN.py:138-150 / :331-336), see csrc/collapse_rows.hip"""

    @staticmethod
    def _block(x, w, tr=None):
        B, C, T, V = x.shape
        t = _lib.RowsConv()
        t.B, t.C, t.T, t.V, t.O = B, C, T, V, w.shape[0]
        t.x, t.W = x.data_ptr(), w.data_ptr()
        if tr is not None:                                   # PReLU(BatchNorm(x)) applied on load (`tower_maps(..., defer=True)`)
            bn = tr["bn"]
            tr["consumed"] = True                            # the consumer fills `save`: the producer's backward checks that there was one
            t.in_on, t.in_train = 1, 1 if tr["train"] else 0
            t.in_bn.stats, t.in_bn.gamma, t.in_bn.beta = _ptr(tr["stats"]), tr["gamma"].data_ptr(), tr["beta"].data_ptr()
            t.in_bn.running_mean, t.in_bn.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            t.in_bn.num_batches_tracked = bn.num_batches_tracked.data_ptr()
            t.in_bn.momentum, t.in_bn.eps, t.in_bn.save = bn.momentum, bn.eps, tr["save"].data_ptr()
            t.in_alpha = tr["alpha"].data_ptr()
            if tr.get("want_tap") and "tap" not in tr:           # forward only: the activated input, for branch records (tests)
                tr["tap"] = torch.empty_like(x)
                t.in_tap = tr["tap"].data_ptr()
        return t

    @staticmethod
    def forward(ctx, want_stats, tr, x, w):
        ctx.set_materialize_grads(False)
        _chk(x)
        w = w if w.is_contiguous() else _copy(w)
        B, C, T, V = x.shape
        O = w.shape[0]
        t = _CollapseRows._block(x, w, tr)
        ctx.tr = tr
        y = torch.empty(B, O, V, dtype=torch.float32, device=x.device)
        stats = _arena(x.device).take(2 * O * _lib.STAT_REPLICAS) if want_stats else None
        t.y, t.stats = y.data_ptr(), _ptr(stats)
        _lib.call("cg_collapse_rows_fwd", ctypes.byref(t), _stream(x))
        ctx.save_for_backward(x, w)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _=None):
        x, w = ctx.saved_tensors
        if dy is None:
            return None, None, None, None
        dy = dy if dy.is_contiguous() else _copy(dy)
        B, C, T, V = x.shape
        t = _CollapseRows._block(x, w, ctx.tr)
        if ctx.tr is not None and ctx.tr.get("fold_reduce", _FOLD_REDUCE):
            # the reduction pass of the BatchNorm / PReLU backward (sums of g, g * xhat, slope gradient) is done here, where dx' is produced
            red = _arena(x.device).take(2 * C + _lib.ALPHA_SLOTS)
            t.in_red = red.data_ptr()
            ctx.tr["red"] = red
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        zb, _z = _zeros(int(_lib.lib().cg_collapse_rows_ws_floats(C, T, w.shape[0])), x.device)
        t.dy, t.dx, t.dW, t.ws = dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), zb.data_ptr()
        _lib.call("cg_collapse_rows_bwd", ctypes.byref(t), _stream(x))
        return None, None, dx if ctx.needs_input_grad[2] else None, dw if ctx.needs_input_grad[3] else None


def collapse_rows(x, w, want_stats=False, transform=None):
    """(y (B,O,V), f64 channel sums or None) of the frame-collapsing convolution w (O,C,T) of x (B,C,T,V); `transform`: x is a raw map of
    `tower_maps(..., defer=True)`, its BatchNorm + PReLU are applied on load."""
    return _CollapseRows.apply(bool(want_stats), transform, x, w)


def collapse_cols_ok(x, w):
    """True when `collapse_cols` takes the (1,V) convolution `w` (O,C,V) of x (B,C,T,V)."""
    if not _ROWS_KERNELS or x.dim() != 4 or not x.is_contiguous() or w.dim() != 3:
        return False
    B, C, T, V = x.shape
    return w.shape[1] == C and w.shape[2] == V and T <= 64 and w.shape[0] <= 64 and (C * V) % 4 == 0 and C * T * V < 2 ** 31


class _CollapseCols(torch.autograd.Function):
    """y[b,o,t] = sum_{c,v} w[o,c,v] x[b,c,t,v] (nn.Conv2d(C, O, (1,V)), CISTGCN.py:152-163), see csrc/collapse_rows.hip"""

    @staticmethod
    def forward(ctx, want_stats, tr, x, w):
        ctx.set_materialize_grads(False)
        _chk(x)
        w = w if w.is_contiguous() else _copy(w)
        B, C, T, V = x.shape
        O = w.shape[0]
        t = _CollapseRows._block(x, w, tr)
        ctx.tr = tr
        y = torch.empty(B, O, T, dtype=torch.float32, device=x.device)
        stats = _arena(x.device).take(2 * O * _lib.STAT_REPLICAS) if want_stats else None
        t.y, t.stats = y.data_ptr(), _ptr(stats)
        _lib.call("cg_collapse_cols_fwd", ctypes.byref(t), _stream(x))
        ctx.save_for_backward(x, w)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _=None):
        x, w = ctx.saved_tensors
        if dy is None:
            return None, None, None, None
        dy = dy if dy.is_contiguous() else _copy(dy)
        B, C, T, V = x.shape
        t = _CollapseRows._block(x, w, ctx.tr)
        if ctx.tr is not None and ctx.tr.get("fold_reduce", _FOLD_REDUCE):
            # the reduction pass of the BatchNorm / PReLU backward (sums of g, g * xhat, slope gradient) is done here, where dx' is produced
            red = _arena(x.device).take(2 * C + _lib.ALPHA_SLOTS)
            t.in_red = red.data_ptr()
            ctx.tr["red"] = red
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        zb, _z = _zeros(int(_lib.lib().cg_collapse_cols_ws_floats(C, V, w.shape[0])), x.device)
        t.dy, t.dx, t.dW, t.ws = dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), zb.data_ptr()
        _lib.call("cg_collapse_cols_bwd", ctypes.byref(t), _stream(x))
        return None, None, dx if ctx.needs_input_grad[2] else None, dw if ctx.needs_input_grad[3] else None


def collapse_cols(x, w, want_stats=False, transform=None):
    """(y (B,O,T), f64 channel sums or None) of the joint-collapsing convolution w (O,C,V) of x (B,C,T,V); `transform` as in `collapse_rows`."""
    return _CollapseCols.apply(bool(want_stats), transform, x, w)
