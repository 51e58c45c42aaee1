"""Evaluation-harness counterparts around the model; the joint gather / scatter and the per-frame
MPJPE kernel live in `ops.gather_joints` / `ops.eval_scatter_mpjpe`.

* `capture_interpretation`: `environment/test.py:146-157` - after a forward, every dotted key of
  `evaluation_config.interpretation.layers` (predict.yaml:162-197) is resolved attribute by attribute on the model and
  appended as `squeeze().cpu().numpy()`; a key the model does not have prints "<key> is not available on model" and is
  skipped (model sizes differ in the number of blocks).
* `save_interpretation`: the `.npy` dictionary the reference's figure scripts read back with `np.load(..., allow_pickle=True).all()`
  (figures_temp.py:54-69): {action: {"interpretation": {key: [arrays]}}}.
* `mpjpe_ms_table`: the per-horizon line of `train.py:40-43`: frames are 40 ms apart, indices [1, 4, 9, 13, 17, 24] of a
  25-frame prediction (80 ... 1000 ms), [1, 4, 9] of a 10-frame one.
"""
import numpy as np


def capture_interpretation(model, interpretation_keywords, store=None):
    store = {} if store is None else store
    for key in interpretation_keywords or ():
        try:
            obj = model
            for part in key.split("."):
                obj = getattr(obj, part)
            value = obj.detach().squeeze().cpu().numpy()
        except Exception:
            print("%s is not available on model" % key)
            continue
        store.setdefault(key, []).append(value)
    return store


def save_interpretation(path, store, action="all"):
    np.save(path, {action: {"interpretation": store}}, allow_pickle=True)
    return path


def mpjpe_ms_table(mpjpe_seq):
    """(values at the reported horizons as {ms: error}, the reference's printed line)"""
    v = np.asarray(mpjpe_seq, dtype=np.float64).reshape(-1)
    idx = [1, 4, 9, 13, 17, 24] if len(v) > 10 else [1, 4, 9]
    idx = [i for i in idx if i < len(v)]
    cells = ["%d:%.2f," % (40 * (i + 1), v[i]) for i in idx]
    return {40 * (i + 1): float(v[i]) for i in idx}, "mpjpe: " + " ".join(cells)
