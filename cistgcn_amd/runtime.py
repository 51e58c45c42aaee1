"""Step runtime around the model: HIP-graph capture of forward + loss + backward, the flat gradient
buffer, and data-parallel gradient averaging over RCCL (one process per GPU).

The reference has no distributed code at all (train.py:214 pins one GPU); batches are independent
except for per-replica BatchNorm statistics, so data parallelism = every rank runs the same step
on its shard and the flat gradient buffer is all-reduced (sum) and scaled by 1/world.
"""
import os
import ctypes

import numpy as np
import torch

from . import _lib, ops

_CHUNK = 2048


class FlatGrads:
    """One contiguous fp32 buffer holding all parameter gradients, filled by a single gather kernel.  The buffer can be
    cut into contiguous BUCKETS of whole tensors (`set_buckets`), gathered one at a time: the data-parallel step
    all-reduces the bucket of the layers whose gradients land first while the rest of the backward pass still runs."""

    def __init__(self, params, device):
        self.params = [p for p in params if p.requires_grad]
        sizes = [p.numel() for p in self.params]
        # every tensor starts on a 16-byte boundary of the flat buffer (slots padded to 4 floats; the padding stays zero):
        # parameters re-homed into the same layout (FlatAdam) keep the aligned float4 launch plans
        self.offsets = np.concatenate([[0], np.cumsum([(n + 3) & ~3 for n in sizes])]).astype(np.int64)
        self.sizes = sizes
        self.numel = int(self.offsets[-1])
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        ct, cb, cl, first = [], [], [], []
        for t, n in enumerate(sizes):
            first.append(len(ct))
            for b in range(0, n, _CHUNK):
                ct.append(t); cb.append(b); cl.append(min(_CHUNK, n - b))
        first.append(len(ct))
        self.n_chunks = len(ct)
        self._first_chunk = first                 # tensor index -> index of its first chunk
        dev = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).to(device)
        self._off = dev(self.offsets[:-1], np.int64)
        self._ct, self._cb, self._cl = dev(ct, np.int32), dev(cb, np.int32), dev(cl, np.int32)
        # one pointer table per bucket (key None = the whole buffer): buckets are gathered on different streams in the two-phase
        # data-parallel step, and a shared table would be rewritten for one bucket while the gather kernel of another still
        # reads it
        self._tables = {}
        self._device = device
        self.buckets = [(0, len(sizes))]          # tensor index ranges

    def set_buckets(self, tensor_cuts):
        """Cut the buffer at the given tensor indices, e.g. [k] -> buckets [0,k) and [k,n)."""
        cuts = [0] + sorted(int(c) for c in tensor_cuts if 0 < int(c) < len(self.params)) + [len(self.params)]
        self.buckets = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        return self.buckets

    def bucket_view(self, k):
        a, b = self.buckets[k]
        return self.flat[int(self.offsets[a]):int(self.offsets[b])]

    def _copy(self, direction, bucket=None, scale=1.0):
        a, b = self.buckets[bucket] if bucket is not None else (0, len(self.params))
        ptrs = []
        for i, p in enumerate(self.params):
            if p.grad is None:
                if a <= i < b:
                    raise RuntimeError("FlatGrads: parameter %d has no gradient" % i)
                ptrs.append(0)                   # outside this bucket (its gradient lands in a later phase): not touched
                continue
            if not p.grad.is_contiguous():
                raise RuntimeError("FlatGrads: non-contiguous gradient")
            ptrs.append(p.grad.data_ptr())
        tab = self._tables.get(bucket)
        if tab is None:
            host = torch.zeros(len(self.params), dtype=torch.int64)
            if torch.device(self._device).type == "cuda":
                host = host.pin_memory()
            tab = self._tables[bucket] = {"host": host, "dev": torch.zeros(len(self.params), dtype=torch.int64, device=self._device), "last": None}
        if ptrs != tab["last"]:
            # the pointer table only changes when autograd hands out new gradient buffers; under HIP-graph replay
            # the buffers are static and this upload happens once
            tab["host"].copy_(torch.tensor(ptrs, dtype=torch.int64))
            tab["dev"].copy_(tab["host"])                # synchronous: the host staging buffer is reused
            tab["last"] = ptrs
        c0, c1 = self._first_chunk[a], self._first_chunk[b]
        if c1 <= c0:
            return
        _lib.call("cg_multi_copy", ops._ptr(tab["dev"]), ops._ptr(self._off), ops._ptr(self._ct[c0:]), ops._ptr(self._cb[c0:]),
                  ops._ptr(self._cl[c0:]), c1 - c0, ops._ptr(self.flat), direction, float(scale), ops._stream(self.flat))

    def gather(self, bucket=None, scale=1.0):
        """p.grad * scale -> flat (one launch); `bucket` restricts it to one bucket of tensors."""
        self._copy(0, bucket, scale)
        return self.flat if bucket is None else self.bucket_view(bucket)

    def scatter(self, bucket=None):
        """flat -> p.grad (one launch)."""
        self._copy(1, bucket)


class FlatAdam(torch.optim.Optimizer):
    """Adam on ONE flat fp32 buffer.  Semantics of the reference's optimizer,
    `torch.optim.Adam(params, lr, weight_decay)` (environment/utils.py:53-57): L2 decay added to the gradient,
    bias-corrected moments, eps after the square root; optional `clip_grad_value_` (environment/train.py:97-98).
    The parameters are re-homed into slices of one buffer (views, no copies afterwards), gradients arrive through
    `FlatGrads`, so the 698 per-tensor update launches of the stock optimizer become two kernels (gather + update).

    It is a `torch.optim.Optimizer` with one parameter group, so the reference's schedulers drive it unchanged:
    `LearningRateWarmUP` writes `param_groups[0]['lr']` (environment/utils.py:6-27) and `lr_scheduler.StepLR /
    MultiStepLR / CosineAnnealingLR` (:30-43) accept it; the learning rate is read from the group at every step."""

    def __init__(self, model, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0, flat=None):
        params = [p for p in (model.parameters() if isinstance(model, torch.nn.Module) else model) if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_value=clip_value))
        self.params = params
        device = params[0].device
        self.grads = flat if flat is not None else FlatGrads(params, device)
        self.flat_param = torch.zeros(self.grads.numel, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, off in zip(params, self.grads.offsets[:-1]):
                n, off = p.numel(), int(off)
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + n].view(p.shape)       # same values, new (16-byte aligned) home
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.step_count = 0

    # the hyper-parameters live in the (single) parameter group, where schedulers and callers expect them
    lr = property(lambda self: self.param_groups[0]["lr"], lambda self, v: self.param_groups[0].__setitem__("lr", v))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, gathered=False):
        """One update.  `grad_scale` folds the 1/world of a data-parallel sum; `gathered` = the flat gradient
        buffer is already filled (e.g. by GraphedStep / after the all-reduce)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not gathered:
            self.grads.gather()
        self.step_count += 1
        g, f = self.param_groups[0], self.flat_param
        _lib.call("cg_adam_flat", ops._ptr(f), ops._ptr(self.grads.flat), ops._ptr(self.exp_avg), ops._ptr(self.exp_avg_sq),
                  f.numel(), float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], float(g["weight_decay"]), grad_scale,
                  float(g["clip_value"]), self.step_count, ops._stream(f))
        return loss

    def state_dict(self):
        """The checkpoint 'optimizer' entry (train.py:184-194) in the layout of the reference's own optimizer,
        `torch.optim.Adam.state_dict()`: per-parameter `step / exp_avg / exp_avg_sq` + one parameter group, so that a
        checkpoint written here resumes in the reference (`model_loader.py:23`) and vice versa."""
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group.pop("clip_value", None)
        for k, v in (("amsgrad", False), ("maximize", False), ("foreach", None), ("capturable", False), ("differentiable", False),
                     ("fused", None)):
            group.setdefault(k, v)
        group["params"] = list(range(len(self.params)))
        state = {}
        if self.step_count > 0:
            for i, (p, off) in enumerate(zip(self.params, self.grads.offsets[:-1])):
                n, off = p.numel(), int(off)
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        return {"state": state, "param_groups": [group], "clip_value": self.param_groups[0].get("clip_value", 0.0)}

    def load_state_dict(self, state):
        """Accepts a `torch.optim.Adam` state dict (what the reference stores in `ckpt['optimizer']`, and what
        `state_dict()` above writes) or the flat layout of earlier builds."""
        if state.get("flat"):
            if state["exp_avg"].numel() == self.exp_avg.numel():
                self.exp_avg.copy_(state["exp_avg"]); self.exp_avg_sq.copy_(state["exp_avg_sq"])
            else:                                   # written before the slots were padded to 16 bytes: dense layout
                o = 0
                for p, off in zip(self.params, self.grads.offsets[:-1]):
                    n, off = p.numel(), int(off)
                    self.exp_avg[off:off + n].copy_(state["exp_avg"][o:o + n]); self.exp_avg_sq[off:off + n].copy_(state["exp_avg_sq"][o:o + n])
                    o += n
            self.step_count = int(state["step"])
            self.param_groups[0].update(state["param_group"])
            return
        groups = state.get("param_groups")
        if not groups or len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("FlatAdam.load_state_dict: expected one parameter group over %d tensors" % len(self.params))
        if groups[0].get("amsgrad") or groups[0].get("maximize"):
            raise ValueError("FlatAdam.load_state_dict: amsgrad / maximize states are not supported")
        per = state.get("state", {})
        steps = set()
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for i, (p, off) in enumerate(zip(self.params, self.grads.offsets[:-1])):
            st = per.get(i, per.get(str(i)))
            if st is None:
                continue
            n, off = p.numel(), int(off)
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("FlatAdam.load_state_dict: moment %d has shape %s, parameter %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1)); self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: parameters are at different step counts %s" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
        g = self.param_groups[0]
        for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
            if k in groups[0]:
                g[k] = groups[0][k]
        if "clip_value" in state:
            g["clip_value"] = state["clip_value"]


def set_optimizer(model, opt):
    """Counterpart of the reference's `set_optimizer` (environment/utils.py:53-57): Adam with the YAML's `lr` and
    `weight_decay` (`opt.learning_config`), as one flat-buffer update."""
    lc = opt.learning_config
    return FlatAdam(model, lr=float(lc.lr), weight_decay=float(lc.weight_decay))


def allreduce_mean_(flat, group=None):
    """In-place mean over the data-parallel group: RCCL all-reduce(sum) over xGMI (backend "nccl" on ROCm) or gloo in the
    CPU tests, then 1/world on the same buffer (one library kernel).  `DataParallelStep` does not need that kernel: it folds
    the shard weight into the gradient gather and the 1/world into the Adam update."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    _lib.call("cg_scale", ops._ptr(flat), flat.numel(), 1.0 / world, ops._stream(flat))
    return flat


def shard_weights(local_batch, group=None):
    """Weight of this replica's gradient when per-GPU batches differ (BASELINE config 5): B_r * world / sum B, so that
    the plain mean of the weighted replica gradients is the gradient of the global mean loss."""
    import torch.distributed as dist
    t = torch.tensor([float(local_batch)])
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    total = t.clone()
    dist.all_reduce(total, group=group)
    return float(local_batch) * dist.get_world_size(group) / float(total.item())


_GRAPH_ATTRS = ("Adj", "w1", "w2", "joints", "displacements", "seq_joints", "seq_joints_n", "seq_joints_dims")


def _drop_graph_attributes(model):
    """The interpretation attributes are un-detached graph tensors (as in the reference, CISTGCN.py:262,381-382,
    469-473); forget the ones of a finished pass so that its autograd graph is released."""
    for m in model.modules():
        for k in _GRAPH_ATTRS:
            if k in m.__dict__:
                del m.__dict__[k]


def _buffers_snapshot(model, device):
    """BatchNorm running statistics / counters and the dropout seed: warm-up and probe passes must not move them."""
    bufs = [b for b in model.buffers()]
    return [b.clone() for b in bufs], ops.seed_state(device).clone()


def _buffers_restore(model, device, snap):
    with torch.no_grad():
        for b, v in zip(model.buffers(), snap[0]):
            b.copy_(v)
        ops.seed_state(device).copy_(snap[1])


class _StepBase:
    """Shared by the step objects: private zero pool (installed only around their own passes), static inputs, guard that
    the parameters still live where the captured launches read them."""

    def _init_common(self, model, x, target, flat):
        self.model, self.flat = model, flat
        self.x, self.target = x, target
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._pool = ops.new_step_pool(x.device)
        self._one = torch.ones((), dtype=torch.float32, device=x.device)
        self._param_ptrs = None

    def _pass(self):
        """forward + MPJPE + backward of the static batch"""
        for p in self.params:
            p.grad = None
        pred, = self.model(self.x)
        loss = ops.mpjpe(pred, self.target)
        loss.backward(self._one)            # explicit root gradient: autograd would fill a fresh ones tensor with a stock kernel
        return loss

    def _freeze_param_pointers(self):
        self._param_ptrs = [p.data_ptr() for p in self.params]

    def _check_param_pointers(self):
        if self._param_ptrs is not None and any(p.data_ptr() != q for p, q in zip(self.params, self._param_ptrs)):
            raise RuntimeError("the parameters moved after this step was captured (FlatAdam / model.to() / load with assign): "
                               "the HIP graph still reads the old buffers. Build FlatAdam BEFORE the step object.")


class GraphedStep(_StepBase):
    """forward + MPJPE + backward of one batch, captured once in a HIP graph and replayed: the ~2 k
    kernel launches of a step collapse into one graph launch, which is what makes the B=16
    configuration viable.  Inputs live in static device buffers (`x`, `target`).  With `flat`, the
    gradients (static buffers under replay) are gathered into the flat all-reduce buffer by one
    kernel launched right after the graph.

    Order matters and is enforced: anything that re-homes the parameters (`FlatAdam(model)`, `model.to()`) must happen
    BEFORE the capture; `replay()` raises if a parameter has moved since.  Construction leaves the model state untouched:
    BatchNorm running statistics, `num_batches_tracked` and the dropout seed are restored after warm-up and probing."""

    def __init__(self, model, x, target, warmup=3, flat=None, branches=False, tries=1):
        """tries > 1: capture that many graphs (each lands in different memory), time a few replays of each and keep the
        median one (the same step replays in 5.27 .. 5.46 ms depending on where the capture's buffers were placed, stable
        for the life of a capture, profiles/README.md); `capture_ms` lists all probes."""
        self._init_common(model, x.clone(), target.clone(), flat)
        # Optional: the context branch on a forked stream (one fork / join per step, CISTGCN._parallel) -> a parallel graph
        # branch.  Captures and replays correctly on ROCm 7.2 but measured no faster (5.44 vs 5.43 ms at B=16), so it is off.
        model.branch_streams = bool(branches)
        snap = _buffers_snapshot(model, x.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _drop_graph_attributes(model)      # Adj / w1 / ... of the warm-up pass keep its autograd graph (and streams) alive
        cands = []
        for _ in range(max(1, int(tries))):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                loss = self._step()
            # keep the VALUE only: the tensor with its grad_fn would keep this capture's AccumulateGrad nodes (bound to this
            # capture's stream) alive, and the next capture would run backward across mismatched streams
            loss = loss.detach()
            cand = [0.0, graph, loss, [p.grad for p in self.params]]
            if tries > 1:
                cand[0] = self._probe(graph)
            _drop_graph_attributes(model)
            cands.append(cand)                 # all kept alive until the end so that the next capture lands elsewhere
        self.capture_ms = [round(c[0], 4) for c in cands] if tries > 1 else None
        best = sorted(cands, key=lambda c: c[0])[len(cands) // 2]      # the median capture: a representative, not the best
        del cands
        self.graph, self.loss = best[1], best[2]
        for p, g in zip(self.params, best[3]):     # the parameters' .grad must be the buffers THIS graph writes
            p.grad = g
        _buffers_restore(model, x.device, snap)
        self._freeze_param_pointers()

    @staticmethod
    def _probe(graph, n=20):
        import time
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def _step(self):
        with ops.step_pool(self._pool):
            return self._pass()

    def replay(self):
        self._check_param_pointers()
        self.graph.replay()
        if self.flat is not None:
            self.flat.gather()
        return self.loss


class GraphedForward:
    """Eval-mode forward of one static batch as a HIP graph (inference / validation path, `environment/test.py:279-350`
    of the reference calls the model under `no_grad` like this)."""

    def __init__(self, model, x, warmup=2):
        self.model = model
        self.x = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                model(self.x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _drop_graph_attributes(model)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.pred, = model(self.x)

    def replay(self):
        self.graph.replay()
        return self.pred


class EagerStep(_StepBase):
    """Same step without graph capture (debugging / first-iteration reference)."""

    def __init__(self, model, x, target, flat=None):
        self._init_common(model, x, target, flat)

    def replay(self):
        with ops.step_pool(self._pool):
            self.loss = self._pass()
        if self.flat is not None:
            self.flat.gather()
        return self.loss


class DataParallelStep(_StepBase):
    """One data-parallel training step of this replica (BASELINE configs[3] and [4]): forward + MPJPE +
    backward on the local shard, gradient mean over the group, optional FlatAdam update.

    * Unequal per-GPU batches: each replica's gradient is weighted B_r * world / sum B (`shard_weights`) inside the gather
      kernel, the 1/world goes into the Adam kernel's `grad_scale` (or one `cg_scale` without optimizer): the result is the
      gradient of the global mean loss, with per-replica BatchNorm statistics as in the (single-GPU) reference.
    * Overlap: the autograd graph is cut behind input block `cut_block`.  Phase 1 = forward + backward down to the cut
      (output block, context branch, time extrapolator, upper input blocks: their gradients are the tail of the flat
      buffer); its bucket is gathered and all-reduced on a side stream while phase 2 (the remaining input blocks) runs.
      With `graph=True` the two phases are two HIP graphs that share one memory pool.
    The only collective is the all-reduce (RCCL over xGMI under backend "nccl"; gloo in the CPU tests)."""

    def __init__(self, model, x, target, optimizer=None, group=None, cut_block=None, graph=True, warmup=3, flat=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.optimizer = optimizer
        flat = optimizer.grads if optimizer is not None else (flat if flat is not None else FlatGrads(model.parameters(), x.device))
        self._init_common(model, x.clone() if graph else x, target.clone() if graph else target, flat)
        self.weight = shard_weights(x.shape[0], group) if self.world > 1 else 1.0
        nblk = len(model.st_gcnns)
        self.cut_block = (nblk // 2 - 1 if cut_block is None else int(cut_block))
        self.two_phase = 0 <= self.cut_block < nblk - 1
        if self.two_phase:
            first_late = {id(p) for b in list(model.st_gcnns)[:self.cut_block + 1] for p in b.parameters()}
            k = sum(1 for p in flat.params if id(p) in first_late)
            if [id(p) in first_late for p in flat.params] != [True] * k + [False] * (len(flat.params) - k):
                raise RuntimeError("DataParallelStep: the late-bucket parameters are not a prefix of the flat buffer")
            flat.set_buckets([k])                # bucket 0 = blocks [0, cut] (gradients land last), bucket 1 = the rest
        self._cut_in = self._cut_leaf = None
        self.graphs = None
        self._side, self.overlap_probe = None, None
        self.force_collective = False     # run the all-reduce in a one-rank group too (overlap measurements on one GPU)
        if x.is_cuda:
            self.pick_side_stream(collective=self.world > 1)
        self.record_events = False        # replay() then leaves HIP events of both streams in self.events
        self.events = None
        if graph:
            self._capture(warmup)
        self._freeze_param_pointers()

    def pick_side_stream(self, collective=False, tries=8, new_groups=3):
        """HIP maps its streams onto a few hardware queues, and two streams on the same queue run one after the other whatever
        the program says (measured on the MI355X box: about one pool stream in eight delays the default stream,
        tools/diag_stream_overlap.py; the stream the nccl backend launches its collectives on can be such a stream, and then
        every all-reduce - behind whatever it waits for - sits in front of the main stream's next kernel).  The side stream of
        the early bucket is therefore CHOSEN: a ~0.3 ms spin kernel goes on a candidate (with `collective`, followed by a
        one-element all-reduce of the group, whose own stream then waits behind the spin), an event on the main stream says
        whether the main stream ran beside it.  When no candidate passes with the collective in the chain, the group itself is
        replaced (`dist.new_group` over the same ranks gets another stream for its collectives; all ranks decide together) up
        to `new_groups` times.  `overlap_probe` keeps the evidence: {"independent", "collective", "main_ms", "side_ms",
        "tried", "groups"}."""
        import torch.distributed as dist
        dev = self.x.device
        if os.environ.get("CISTGCN_SIDE_STREAM_PROBE", "1") == "0":      # tuning aid: take the next pool stream unmeasured
            self._side, self.overlap_probe = torch.cuda.Stream(device=dev), {"independent": None, "collective": False, "tried": 0, "groups": 0}
            return self.overlap_probe
        main = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200000)
        e0.record(); torch.cuda._sleep(200000); e1.record(); torch.cuda.synchronize()
        spin = int(200000 * 0.3 / max(e0.elapsed_time(e1), 1e-3))
        collective = bool(collective and dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl")
        tiny = torch.zeros(1, device=dev) if collective else None
        kept, last = [], None

        def probe(group):
            nonlocal last
            if collective:
                dist.all_reduce(tiny, group=group)              # communicator and its stream exist before the probe
            for n in range(1, tries + 1):
                cand = torch.cuda.Stream(device=dev)
                kept.append(cand)                               # rejected candidates stay alive: the next one is another stream
                torch.cuda.synchronize()
                a, b, c = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                a.record()
                cand.wait_stream(main)
                with torch.cuda.stream(cand):
                    torch.cuda._sleep(spin)
                    if collective:
                        dist.all_reduce(tiny, group=group, async_op=True).wait()
                    c.record()
                b.record()
                torch.cuda.synchronize()
                last = {"independent": a.elapsed_time(b) < 0.5 * a.elapsed_time(c), "collective": collective,
                        "main_ms": round(a.elapsed_time(b), 4), "side_ms": round(a.elapsed_time(c), 4), "tried": n}
                if last["independent"]:
                    return True
            return False

        last = self._choose_group(probe, collective, new_groups, dev, lambda: last)
        self._side, self.overlap_probe = kept[-1], last
        return last

    def _choose_group(self, probe, collective, new_groups, dev, result, backend="nccl"):
        """The regrouping loop of `pick_side_stream` (separate so that a test can drive it with a stubbed probe over gloo):
        probe the group; when a rank found no independent stream with the collective in the chain, ALL ranks replace the
        group together.  `dist.new_group` must be entered by every rank of the default group, so only a step that runs on the
        default group (group None / WORLD) regroups - a step on a sub-group keeps its group and says so in the record.  Whether
        the replacement worked is agreed by a second MIN all-reduce (one rank's exception must not make the ranks diverge);
        a replaced group that this object created is destroyed."""
        import torch.distributed as dist
        groups, made = 0, None
        while True:
            ok = probe(self.group)
            last = result()
            if not collective:
                break
            flag = torch.tensor([1.0 if ok else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)        # every rank has to have found one
            if float(flag.item()) > 0.0 or groups >= new_groups:
                break
            if self.group is not None and self.group is not dist.group.WORLD and self.group is not made:
                last["regroup_skipped"] = "the step runs on a sub-group: new_group needs every rank of the default group"
                break
            err, fresh = None, None
            try:
                ranks = dist.get_process_group_ranks(self.group if self.group is not None else dist.group.WORLD)
                fresh = dist.new_group(ranks=ranks, backend=backend)
            except (RuntimeError, ValueError) as e:             # keep the group: the step is correct either way, only not overlapped
                err = str(e)[:200]
            agreed = torch.tensor([0.0 if err else 1.0], device=dev)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN, group=self.group)      # the outcome of new_group, agreed by all ranks
            if float(agreed.item()) <= 0.0:
                if fresh is not None:
                    dist.destroy_process_group(fresh)
                last["new_group_error"] = err or "another rank could not create the group"
                break
            if made is not None:
                dist.destroy_process_group(made)               # only groups this loop created; the caller's group is the caller's
            self.group = made = fresh
            groups += 1
        last["groups"] = groups
        return last

    # ---- the two phases ---------------------------------------------------------------------------------------
    def _cut(self, h):
        t, st = h if isinstance(h, tuple) else (h, None)
        self._cut_in = t
        self._cut_leaf = t.detach().requires_grad_(True)
        return (self._cut_leaf, st) if st is not None else self._cut_leaf

    def _phase1(self):
        self.model.backward_cut = (self.cut_block, self._cut) if self.two_phase else None
        try:
            return self._pass()
        finally:
            self.model.backward_cut = None

    def _phase2(self):
        if self.two_phase:
            self._cut_in.backward(self._cut_leaf.grad)

    def _capture(self, warmup):
        snap = _buffers_snapshot(self.model, self.x.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), ops.step_pool(self._pool):
            for _ in range(warmup):
                self._phase1(); self._phase2()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _drop_graph_attributes(self.model)
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with ops.step_pool(self._pool):
            with torch.cuda.graph(g1):
                loss = self._phase1()
            if self.two_phase:
                with torch.cuda.graph(g2, pool=g1.pool()):
                    self._phase2()
        self.loss = loss.detach()
        self._cut_in = self._cut_leaf = None
        _drop_graph_attributes(self.model)
        self.graphs = (g1, g2 if self.two_phase else None)
        _buffers_restore(self.model, self.x.device, snap)

    # ---- one step ---------------------------------------------------------------------------------------------
    def _reduce(self, bucket, async_op):
        import torch.distributed as dist
        buf = self.flat.gather(bucket, scale=self.weight)
        if self.world > 1 or (self.force_collective and dist.is_initialized()):
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return None

    def replay(self):
        self._check_param_pointers()
        cuda = self.x.is_cuda
        if self.graphs is not None:
            self.graphs[0].replay()
        else:
            with ops.step_pool(self._pool):
                self.loss = self._phase1()
        work = None
        ev = None
        if self.record_events and cuda:
            ev = self.events = {k: torch.cuda.Event(enable_timing=True) for k in ("phase1_end", "reduce1_start", "reduce1_end", "phase2_start", "phase2_end")}
            ev["phase1_end"].record()
        if self.two_phase:
            if cuda:                                   # early bucket: gather + all-reduce beside phase 2
                self._side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._side):
                    if ev:
                        ev["reduce1_start"].record()
                    work = self._reduce(1, True)
                    if ev:
                        if work is not None:
                            work.wait()                # stream-level wait: orders the end event behind the collective
                        ev["reduce1_end"].record()
            else:
                work = self._reduce(1, True)
            if ev:
                ev["phase2_start"].record()
            if self.graphs is not None:
                self.graphs[1].replay()
            else:
                with ops.step_pool(self._pool):
                    self._phase2()
                self._cut_in = self._cut_leaf = None
            if ev:
                ev["phase2_end"].record()
            self._reduce(0, False)
            if work is not None:
                work.wait()
            if cuda:
                torch.cuda.current_stream().wait_stream(self._side)
        else:
            self._reduce(None, False)
        if self.optimizer is not None:
            self.optimizer.step(grad_scale=1.0 / self.world, gathered=True)
        elif self.world > 1:
            _lib.call("cg_scale", ops._ptr(self.flat.flat), self.flat.flat.numel(), 1.0 / self.world, ops._stream(self.flat.flat))
        return self.loss
