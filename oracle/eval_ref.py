"""TEST INFRASTRUCTURE ONLY (CPU restatement, never imported by the product package).

Evaluation-harness post-processing of the reference, restated with stock torch:
  * `_predict` (environment/test.py:97-132): the model sees `inputs[:, :, dim_used]`; its output is scattered back into
    a copy of the full-skeleton target, then the repeated joints are copied (`dim_repeat_32 <- dim_repeat_22`,
    loaders/h36m_motion_3d.py:55-56);
  * `losses.mpjpe` (losses/losses.py:50-61) with `reduce_axis=(0, 2)`: mean over samples and joints of the joint distance,
    one value per predicted frame (what `Evaluator.compute` accumulates, test.py:71).
Pinned by tests/golden/eval_h36m.npz, written by tools/gen_golden_eval.py from the reference's own `_predict` and `mpjpe`.
"""
import torch


def gather_used(inputs, dim_used):
    return inputs[:, :, list(dim_used)]


def scatter_prediction(outputs, target, dim_used, dim_repeat_32=(), dim_repeat_22=()):
    full = target.clone()
    full[:, :, list(dim_used), :] = outputs
    if len(dim_repeat_32):
        full[:, :, list(dim_repeat_32), :] = outputs[:, :, list(dim_repeat_22), :]
    return full


def mpjpe_frames(predicted, target):
    return torch.norm(predicted - target, 2, dim=-1).mean((0, 2))
