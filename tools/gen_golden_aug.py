#!/usr/bin/env python3
"""Golden vectors for the on-device input pipeline, from the REAL reference: the augmentation chain
`loaders/loader.py::get_transformations` builds from the shipped YAML section (`config/CISTGCN/train_h36m.yaml:45-80`),
applied by `H36m_Motion3D.__getitem__` (`loaders/h36m_motion_3d.py:94-108`), with every `np.random.uniform` draw
recorded so that the device path can be driven by the same numbers.  Build container only; writes data only
(tests/golden/aug_h36m.npz)."""
import importlib, os, sys, types
import numpy as np
import torch

sys.dont_write_bytecode = True
pkg = types.ModuleType("human_motion_prediction")
pkg.__path__ = ["/root/reference/human_motion_prediction"]
sys.modules["human_motion_prediction"] = pkg
for sub in ("utils", "environment", "loaders"):          # package __init__ files pull in plotting / fvcore: stub them
    m = types.ModuleType("human_motion_prediction." + sub)
    m.__path__ = ["/root/reference/human_motion_prediction/" + sub]
    sys.modules["human_motion_prediction." + sub] = m
trs = importlib.import_module("human_motion_prediction.environment.custom_transforms")
from types import SimpleNamespace as NS

# the `augmentations:` section of config/CISTGCN/train_h36m.yaml:45-80, as yaml_utils.Struct would present it
AUG = NS(random_scale=NS(x=[0.95, 1.05], y=[0.90, 1.10], z=[0.95, 1.05]), random_noise="",
         random_flip=NS(x=True, y="", z=True),
         random_rotation=NS(x=[-5, 5], y=[-180, 180], z=[-5, 5]),
         random_translation=NS(x=[-0.10, 0.10], y=[-0.10, 0.10], z=[-0.10, 0.10]))

# loaders/loader.py:42-130 decides the order (flip, rotation, scale, noise, translation); importing loader.py itself needs
# torchvision + every dataset module, so the same constructor calls are issued here in the order of its source lines 47-73
chain = [trs.ToTensor(), trs.RandomFlip(AUG.random_flip.x, AUG.random_flip.y, AUG.random_flip.z),
         trs.RandomRotation(AUG.random_rotation.x, AUG.random_rotation.y, AUG.random_rotation.z),
         trs.RandomScale(AUG.random_scale.x, AUG.random_scale.y, AUG.random_scale.z),
         trs.RandomTranslation(AUG.random_translation.x, AUG.random_translation.y, AUG.random_translation.z)]


def compose(data):
    for t in chain:
        data = t(data)
    return data


ds_mod = importlib.import_module("human_motion_prediction.loaders.h36m_motion_3d")
ds = object.__new__(ds_mod.H36m_Motion3D)               # __getitem__ only needs these three attributes
B, Tin, Tout, J = 12, 10, 25, 22
g = np.random.RandomState(123)
ds.target = (50 + 350 * g.randn(B, Tin + Tout, J, 3)).astype(np.float32)
ds.transform = compose
ds.input_n = Tin

draws = []
orig_uniform = np.random.uniform


def logged(*a, **k):
    v = orig_uniform(*a, **k)
    if isinstance(v, np.ndarray):                 # RandomNoise draws a (joints, 3) array in one call
        draws.extend(float(t) for t in v.reshape(-1))
    else:
        draws.append(float(v))
    return v


np.random.seed(2024)
np.random.uniform = logged
out = {"sample": [], "sample_vel": [], "target": [], "target_vel": [], "target_gvel": [], "processed": [], "ndraws": []}
for i in range(B):
    n0 = len(draws)
    item = ds[i]
    out["ndraws"].append(len(draws) - n0)
    for k in ("sample", "sample_vel", "target", "target_vel", "target_gvel", "processed"):
        out[k].append(np.asarray(item[k], dtype=np.float32))
np.random.uniform = orig_uniform
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "aug_h36m.npz")
np.savez_compressed(path, raw=ds.target, draws=np.array(draws, dtype=np.float64), ndraws=np.array(out["ndraws"]),
                    input_n=np.array(Tin), **{k: np.stack(v) for k, v in out.items() if k != "ndraws"})
print("wrote", os.path.normpath(path), "draws per sample:", out["ndraws"], "shapes:", {k: np.stack(v).shape for k, v in out.items() if k != "ndraws"})

# ---- second fixture: the same chain with RandomNoise (loader.py:62-64, a bare amplitude) and RandomPoseInvers (:121-125, always the
# "h36m" mapping, whose pairs index the 32-joint skeleton) on 32-joint windows
body_utils = importlib.import_module("human_motion_prediction.utils.body_utils")
NOISE = 0.02
chain[:] = [trs.ToTensor(), trs.RandomFlip(AUG.random_flip.x, AUG.random_flip.y, AUG.random_flip.z),
            trs.RandomRotation(AUG.random_rotation.x, AUG.random_rotation.y, AUG.random_rotation.z),
            trs.RandomScale(AUG.random_scale.x, AUG.random_scale.y, AUG.random_scale.z),
            trs.RandomNoise(NOISE),
            trs.RandomTranslation(AUG.random_translation.x, AUG.random_translation.y, AUG.random_translation.z),
            trs.RandomPoseInvers("h36m", 0.5, [], True)]
B, Tin, Tout, J = 10, 10, 25, 32
g = np.random.RandomState(321)
ds.target = (50 + 350 * g.randn(B, Tin + Tout, J, 3)).astype(np.float32)
draws.clear()
np.random.seed(777)
np.random.uniform = logged
out = {"sample": [], "sample_vel": [], "target": [], "target_vel": [], "target_gvel": [], "processed": [], "ndraws": []}
for i in range(B):
    n0 = len(draws)
    item = ds[i]
    out["ndraws"].append(len(draws) - n0)
    for k in ("sample", "sample_vel", "target", "target_vel", "target_gvel", "processed"):
        out[k].append(np.asarray(item[k], dtype=np.float32))
np.random.uniform = orig_uniform
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "aug_h36m_noise_inv.npz")
np.savez_compressed(path, raw=ds.target, draws=np.array(draws, dtype=np.float64), ndraws=np.array(out["ndraws"]), input_n=np.array(Tin),
                    noise=np.array(NOISE), **{k: np.stack(v) for k, v in out.items() if k != "ndraws"})
print("wrote", os.path.normpath(path), "draws per sample:", out["ndraws"])
