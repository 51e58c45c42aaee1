"""On-device input pipeline: the reference augments every sequence on the DataLoader's CPU
workers (scipy `Rotation`, one Python call chain per sample; `num_workers = 0` in the shipped YAMLs,
train_h36m.yaml:86) and copies the finished batch with `.cuda()` (`environment/train.py:57-58`).  Here the raw windows go
to the GPU once, one kernel augments the whole batch and emits the per-item tensors of `H36m_Motion3D.__getitem__`
(`loaders/h36m_motion_3d.py:94-108`), and the next batch's host->device copy runs on a side stream under the current step.

`DeviceAugmentation(opt_trs)` takes the same `learning_config.augmentations` section as `loaders/loader.py::
get_transformations` (:42-130) and keeps its order: flip, rotation, scale, noise, translation, pose inversion.  The random
numbers are drawn on the host with `np.random.uniform` in exactly the order the reference's transform objects would draw them for
the samples of the batch one after the other, so a seeded run reproduces the reference's augmentations; the data-dependent parts
(centroids, extents, the rotation itself) run on the device.  `RandomNoise` (custom_transforms.py:350-400) and `RandomPoseInvers`
(:301-347) are taken in their whole-sequence form; the `seq_idx` / `continuous` variants (unused by the shipped YAML) raise.
"""
import numpy as np
import torch

from .. import _lib, ops

NPAR = 24

# RandomPoseInvers: body_utils.get_reduced_skeleton("h36m", inverse=True) (utils/body_utils.py:167-170) - pairs of the 32-joint
# H3.6M skeleton.  (The reference applies them to whatever tensor it is given: a 22-joint sequence raises IndexError there, and here.)
H36M_INVERSE_PAIRS = ((6, 1), (7, 2), (8, 3), (9, 4), (10, 5), (16, 24), (17, 25), (18, 26), (19, 27), (20, 28), (22, 30), (21, 29), (23, 31))


def _range(v, what):
    """the reference's constructors: a scalar means the degenerate interval [v, v], '' / None / 0 means [0, 0]"""
    if v is None or v == "" or v is False:
        return None
    if isinstance(v, (int, float)):
        return (float(v), float(v))
    v = list(v)
    if len(v) != 2:
        raise ValueError("augmentation %s: expected [low, high], got %r" % (what, v))
    return (float(v[0]), float(v[1]))


def _rotvec_matrix(deg):
    """Rotation matrix of a rotation vector given in degrees (what scipy's `Rotation.from_rotvec(., degrees=True).as_matrix()`
    returns; Rodrigues' formula in f64)."""
    v = np.deg2rad(np.asarray(deg, dtype=np.float64))
    th = float(np.linalg.norm(v))
    if th < 1e-300:
        return np.eye(3)
    k = v / th
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + np.sin(th) * K + (1.0 - np.cos(th)) * (K @ K)


class _Step:
    def __init__(self, kind, ranges, prob_threshold):
        self.kind, self.ranges, self.prob_threshold = kind, ranges, float(prob_threshold)


class DeviceAugmentation:
    def __init__(self, opt_trs=None):
        self.steps = []
        if opt_trs is None:
            return

        def section(name):
            s = getattr(opt_trs, name, None)
            return None if (s is None or s == "") else s

        def variant(s, name):
            for k in ("seq_idx", "continuous"):
                if getattr(s, k, None):
                    raise ValueError("augmentation %s.%s is not supported by the on-device pipeline" % (name, k))
            return float(getattr(s, "prob_threshold", 0.5))

        # loaders/loader.py:46-73 then :74-128 - the order of these blocks is the order of the transforms
        for name in ("random_flip", "random_rotation", "random_scale", "random_noise", "random_translation", "rotation", "scale", "noise",
                     "translation", "flip", "pose_invers"):
            s = section(name)
            if s is None:
                continue
            kind = name.replace("random_", "")
            if name == "random_noise":                     # loader.py:62-64: RandomNoise(opt_trs.random_noise), a bare amplitude
                self.steps.append(_Step("noise", float(s), 0.5))
                continue
            if name == "noise":                            # loader.py:97-103
                self.steps.append(_Step("noise", float(s.noise), variant(s, name)))
                continue
            if name == "pose_invers":                      # loader.py:121-125: always the "h36m" skeleton
                self.steps.append(_Step("invert", H36M_INVERSE_PAIRS, variant(s, name)))
                continue
            if kind == "flip":
                axes = tuple(bool(getattr(s, a, "")) for a in "xyz")
                self.steps.append(_Step("flip", axes, variant(s, name)))
                continue
            r = [_range(getattr(s, a, ""), "%s.%s" % (name, a)) for a in "xyz"]
            if all(v is None for v in r):
                continue                                   # the reference skips a transform whose three entries are ''
            self.steps.append(_Step(kind, tuple(v if v is not None else (0.0, 0.0) for v in r), variant(s, name)))

    @property
    def needs_joints(self):
        return any(st.kind in ("noise", "invert") for st in self.steps)

    def draw(self, batch, rng=None, joints=None):
        """(batch, 24) float32 parameter rows (plus the per-joint noise draws when RandomNoise is configured: then the result is a
        dict {"params", "noise"} and `joints` is required); consumes `rng.uniform` (default: the global `np.random`, which is what
        the reference's transforms use) sample by sample, transform by transform, in the reference's order and only where the
        reference draws (the amounts are drawn only when the coin of that transform says 'apply')."""
        rng = np.random if rng is None else rng
        out = np.zeros((batch, NPAR), dtype=np.float32)
        out[:, 13:16] = 1.0
        has_noise = any(st.kind == "noise" for st in self.steps)
        if has_noise and joints is None:
            raise ValueError("DeviceAugmentation.draw: RandomNoise draws one number per joint and axis: pass joints=")
        noise = np.zeros((batch, int(joints), 3), dtype=np.float32) if has_noise else None
        for b in range(batch):
            # composed parameters of this sample, applied by the kernel in the canonical order flip -> rotate -> scale ->
            # translate; the config order is that order (asserted below), repeated kinds compose only if adjacent steps allow
            seen = []
            for st in self.steps:
                if st.kind in seen:
                    raise ValueError("augmentation '%s' configured twice (random_* and explicit): not supported on device" % st.kind)
                seen.append(st.kind)
                if st.kind == "flip":
                    for a in range(3):
                        if st.ranges[a] and rng.uniform() > st.prob_threshold:
                            out[b, a] = 1.0
                    continue
                if not (rng.uniform() > st.prob_threshold):
                    continue
                if st.kind == "noise":                     # custom_transforms.py:368-370: one array draw of (joints, 3) numbers
                    out[b, 19] = st.ranges
                    noise[b] = np.asarray(rng.uniform(-1, 1, (int(joints), 3)), dtype=np.float64).astype(np.float32)
                    continue
                if st.kind == "invert":
                    out[b, 20] = 1.0
                    continue
                vals = [np.float32(rng.uniform(lo, hi)) for lo, hi in st.ranges]
                if st.kind == "rotation":
                    out[b, 3] = 1.0
                    out[b, 4:13] = _rotvec_matrix(vals).astype(np.float32).reshape(-1)
                elif st.kind == "scale":
                    out[b, 13:16] = vals
                else:
                    out[b, 16:19] = vals
            order = [k for k in ("flip", "rotation", "scale", "noise", "translation", "invert") if k in seen]
            if seen != order:
                raise ValueError("augmentation order %s is not the reference's flip/rotation/scale/noise/translation/inversion order" % seen)
        return {"params": out, "noise": noise} if has_noise else out

    def _perm(self, J, device):
        """joint permutation the reference's sequential pair swaps compose to: output joint j shows joint perm[j]"""
        for st in self.steps:
            if st.kind == "invert":
                idx = list(range(J))
                for x, y in st.ranges:
                    if x >= J or y >= J:
                        raise IndexError("RandomPoseInvers: joint pair (%d, %d) out of range for %d joints (the reference indexes the "
                                         "same way and fails the same way)" % (x, y, J))
                    idx[x], idx[y] = idx[y], idx[x]
                return torch.tensor(idx, dtype=torch.int32, device=device)
        return None

    def __call__(self, raw, input_n, params=None, keep_processed=False):
        """raw: (B, L, J, 3) float32 on the HIP device (un-augmented windows of input_n + output_n frames).  Returns the dict
        of `H36m_Motion3D.__getitem__` / `Amass_Motion3D.__getitem__` (the two are the same code) for the batch: sample, sample_vel,
        target, target_vel, target_gvel (and processed on request; "original" is the caller's `raw`, "item" its indices)."""
        ops._chk(raw, "raw")
        if raw.dim() != 4 or raw.shape[3] != 3:
            raise ValueError("expected windows of shape (B, L, J, 3), got %s" % (tuple(raw.shape),))
        raw = raw if raw.is_contiguous() else ops._copy(raw)
        B, L, J, _ = raw.shape
        if params is None:
            params = self.draw(B, joints=J)
        noise = None
        if isinstance(params, dict):
            params, noise = params["params"], params["noise"]
            if isinstance(noise, np.ndarray):
                noise = torch.from_numpy(np.ascontiguousarray(noise, dtype=np.float32)).to(raw.device, non_blocking=True)
            if tuple(noise.shape) != (B, J, 3):
                raise ValueError("expected (%d, %d, 3) noise draws" % (B, J))
        perm = self._perm(J, raw.device)
        if isinstance(params, np.ndarray):
            params = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float32)).to(raw.device, non_blocking=True)
        if tuple(params.shape) != (B, NPAR):
            raise ValueError("expected a (%d, %d) parameter table" % (B, NPAR))
        dev, f32 = raw.device, torch.float32
        To = L - input_n
        out = {"sample": torch.empty(B, input_n, J, 3, dtype=f32, device=dev), "target": torch.empty(B, To, J, 3, dtype=f32, device=dev),
               "target_vel": torch.empty(B, To, J, 3, dtype=f32, device=dev), "target_gvel": torch.empty(B, To, J, 1, dtype=f32, device=dev),
               "sample_vel": torch.empty(B, input_n, J, 3, dtype=f32, device=dev)}
        proc = torch.empty(B, L, J, 3, dtype=f32, device=dev) if keep_processed else None
        _lib.call("cg_augment_sequences", ops._ptr(raw), ops._ptr(params), ops._ptr(out["sample"]), ops._ptr(out["target"]),
                  ops._ptr(out["target_vel"]), ops._ptr(out["target_gvel"]), ops._ptr(proc), ops._ptr(out["sample_vel"]), ops._ptr(noise), ops._ptr(perm),
                  B, L, J, int(input_n), ops._stream(raw))
        if keep_processed:
            out["processed"] = proc
        return out


class DevicePrefetcher:
    """Iterates (B, L, J, 3) host batches (numpy or CPU tensors) and yields augmented device batches; the host->device copy
    and the augmentation kernel of batch k+1 are issued on a side stream while batch k is being consumed, through two pinned
    staging buffers (`environment/train.py:57-58` copies synchronously inside the hot loop instead)."""

    def __init__(self, batches, augmentation, input_n, device="cuda"):
        self.it, self.aug, self.input_n = iter(batches), augmentation, int(input_n)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._pinned = [None, None]
        self._copied = [None, None]          # event behind the last host->device copy out of each staging buffer
        self._k = 0
        self._next = None
        self._preload()

    def _preload(self):
        try:
            host = next(self.it)
        except StopIteration:
            self._next = None
            return
        host = torch.as_tensor(host, dtype=torch.float32)
        params = self.aug.draw(host.shape[0], joints=host.shape[2])  # host RNG, reference order
        if not self.cuda:
            self._next = self.aug(host.to(self.device), self.input_n, params)
            return
        slot = self._k % 2
        self._k += 1
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()                         # the DMA out of this staging buffer has finished
        if self._pinned[slot] is None or self._pinned[slot].shape != host.shape:
            self._pinned[slot] = torch.empty(host.shape, dtype=torch.float32).pin_memory()
        self._pinned[slot].copy_(host)
        with torch.cuda.stream(self.stream):
            raw = self._pinned[slot].to(self.device, non_blocking=True)
            self._copied[slot] = torch.cuda.Event()
            self._copied[slot].record(self.stream)
            self._next = self.aug(raw, self.input_n, params)
            self._ready = torch.cuda.Event()
            self._ready.record(self.stream)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        batch = self._next
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self._ready)
            for t in batch.values():
                t.record_stream(torch.cuda.current_stream(self.device))
        self._preload()
        return batch
