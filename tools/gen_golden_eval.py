#!/usr/bin/env python3
"""Golden vectors for the evaluation-harness counterpart, from the REAL reference (`environment/test.py::_predict`,
`losses.mpjpe`).  Build container only; writes data only (tests/golden/eval_h36m.npz)."""
import importlib, os, sys, types
import numpy as np
import torch

sys.dont_write_bytecode = True
pkg = types.ModuleType("human_motion_prediction")
pkg.__path__ = ["/root/reference/human_motion_prediction"]
sys.modules["human_motion_prediction"] = pkg
ref_test = importlib.import_module("human_motion_prediction.environment.test")
ref_losses = importlib.import_module("human_motion_prediction.losses.losses")

# the 22 joints the H3.6M loader keeps (all but the ten constant ones), and its repeated joints (h36m_motion_3d.py:55-56)
IGNORED = [0, 1, 6, 11, 16, 20, 23, 24, 28, 31]
DIM_USED = [j for j in range(32) if j not in IGNORED]
REP22, REP32 = [9, 9, 14, 16, 19, 21], [16, 24, 20, 23, 28, 31]


class CISTGCN(torch.nn.Module):          # `_predict` dispatches on the class name (test.py:98-99)
    def __init__(self, canned):
        super().__init__()
        self.canned = canned
        self.seen = None

    def forward(self, x):
        self.seen = x.clone()
        return (self.canned,)


g = torch.Generator().manual_seed(77)
B, Ti, To = 5, 10, 25
inputs = 50 + 350 * torch.randn(B, Ti, 32, 3, generator=g)
target = 50 + 350 * torch.randn(B, To, 32, 3, generator=g)
canned = target[:, :, DIM_USED] + 20 * torch.randn(B, To, 22, 3, generator=g)
loader = types.SimpleNamespace(dataset=types.SimpleNamespace(dim_used=DIM_USED, dim_repeat_22=REP22, dim_repeat_32=REP32))
model = CISTGCN(canned)
with torch.no_grad():
    out = ref_test._predict(model, loader, inputs, target, test_mode=True)
    frames = ref_losses.mpjpe(out, target, reduce_axis=(0, 2))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "eval_h36m.npz")
np.savez_compressed(path, inputs=inputs.numpy(), target=target.numpy(), model_output=canned.numpy(), model_input=model.seen.numpy(),
                    predicted_full=out.numpy(), mpjpe_frames=frames.numpy(), dim_used=np.array(DIM_USED), rep22=np.array(REP22),
                    rep32=np.array(REP32))
print("wrote", os.path.normpath(path), "frames[:3] =", frames[:3].tolist())
