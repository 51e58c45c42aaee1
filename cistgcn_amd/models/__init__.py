"""Drop-in mirror of `human_motion_prediction.models` (reference models/__init__.py:1-2): the two
registry names resolve to the one MI355X implementation."""
from .CISTGCN.CISTGCN import CISTGCN as CISTGCN_0
from .CISTGCN.CISTGCN_eval import CISTGCN as CISTGCN_eval
from .choose_net import choose_net

__all__ = ["CISTGCN_0", "CISTGCN_eval", "choose_net"]
