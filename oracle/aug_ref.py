"""TEST INFRASTRUCTURE ONLY (CPU restatement, never imported by the product package).

The training augmentations and per-item tensors of the reference restated with numpy / stock torch, for batches and
random draws beyond the committed fixture:
  * RandomFlip / RandomRotation / RandomScale / RandomNoise / RandomTranslation / RandomPoseInvers
    (environment/custom_transforms.py:243-298, 10-84, 87-161, 350-400, 164-240, 301-347) without `seq_idx` / `continuous`,
    composed in the order of loaders/loader.py:42-130;
  * `H36m_Motion3D.__getitem__` (loaders/h36m_motion_3d.py:94-108): velocities = frame differences, target_vel /
    target_gvel = cumulative sums from frame input_n - 1 on.
Pinned by tests/golden/aug_h36m.npz and aug_h36m_noise_inv.npz (tools/gen_golden_aug.py runs the reference's own classes with
recorded draws).
"""
import numpy as np
import torch


class Replay:
    """rng stand-in that replays recorded `uniform` draws (the scalar value is returned whatever the bounds)"""

    def __init__(self, draws):
        self.draws, self.i = list(draws), 0

    def uniform(self, *a):
        if len(a) == 3 and a[2] is not None:                      # array draw: uniform(low, high, size) consumes prod(size) numbers
            n = int(np.prod(a[2]))
            v = np.asarray(self.draws[self.i:self.i + n], dtype=np.float64).reshape(a[2])
            self.i += n
            return v
        v = self.draws[self.i]
        self.i += 1
        return v


def rotvec_matrix(deg):
    from scipy.spatial.transform import Rotation as R
    return R.from_rotvec(np.asarray(deg, dtype=np.float64), degrees=True).as_matrix()


def augment_one(data, rng, flip=(True, False, True), rot=((-5, 5), (-180, 180), (-5, 5)), scale=((0.95, 1.05), (0.9, 1.1), (0.95, 1.05)),
                trans=((-0.1, 0.1),) * 3, thr=0.5, noise=None, inverse_pairs=None):
    data = torch.as_tensor(data).clone()
    c = data.mean((0, 1))
    src = data.clone()
    for a in range(3):                                            # custom_transforms.py:264-294
        if flip[a] and rng.uniform() > thr:
            data[:, :, a] = c[a] - (src[:, :, a] - c[a])
    if rng.uniform() > thr:                                       # :50-80
        ang = [np.float32(rng.uniform(lo, hi)) for lo, hi in rot]
        m = torch.from_numpy(rotvec_matrix(ang)).float()
        c = data.mean((0, 1))
        data = torch.matmul(data - c, m) + c
    if rng.uniform() > thr:                                       # :127-157
        s = torch.tensor([np.float32(rng.uniform(lo, hi)) for lo, hi in scale])
        data = data * s
    if noise is not None and rng.uniform() > thr:                 # :367-396 (no seq_idx, not continuous: constant amplitude)
        u = torch.from_numpy(np.asarray(rng.uniform(-1, 1, (data.shape[1], 3)), dtype=np.float64))
        dist = data.max(0).values.max(0).values - data.min(0).values.min(0).values
        data = (data + float(noise) * u * dist).float()
    if rng.uniform() > thr:                                       # :204-236
        t = torch.tensor([np.float32(rng.uniform(lo, hi)) for lo, hi in trans])
        dist = data.max(0).values.max(0).values - data.min(0).values.min(0).values
        data = data + t * dist
    if inverse_pairs is not None and rng.uniform() > thr:         # :323-344 (whole sequence)
        for x, y in inverse_pairs:
            tx, ty = data[:, x, :].clone(), data[:, y, :].clone()
            data[:, x, :] = ty
            data[:, y, :] = tx
    return data


def item_tensors(proc, input_n):
    proc = np.asarray(proc, dtype=np.float32)
    vel = np.diff(proc, axis=0)
    gvel = np.linalg.norm(vel, axis=-1, keepdims=True)
    return {"sample": proc[:input_n], "sample_vel": vel[:input_n], "target": proc[input_n:], "target_vel": vel[input_n - 1:].cumsum(0),
            "target_gvel": gvel[input_n - 1:].cumsum(0), "processed": proc}
