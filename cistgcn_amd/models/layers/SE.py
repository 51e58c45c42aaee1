"""Parameter holders for the squeeze-excite gates (reference models/layers/SE.py:5-41).  The
arithmetic runs in cg_reduce_bc / cg_se_gate / cg_norm_act (see cistgcn_amd/ops.py)."""
import torch.nn as nn


class _SE(nn.Module):
    def __init__(self, channel, hidden):
        super().__init__()
        # slots 1 (ReLU) and 3 (Sigmoid) carry no parameters
        self.excitation = nn.ModuleDict({"0": nn.Linear(channel, hidden, bias=False),
                                         "2": nn.Linear(hidden, channel, bias=False)})

    @property
    def w1(self):
        return self.excitation["0"].weight

    @property
    def w2(self):
        return self.excitation["2"].weight


class SELayer1d(_SE):
    def __init__(self, channel, reduction=4):
        super().__init__(channel, channel // reduction)


class SELayer2d(_SE):
    def __init__(self, channel, reduction=4):
        super().__init__(channel, max(1, channel // reduction))
