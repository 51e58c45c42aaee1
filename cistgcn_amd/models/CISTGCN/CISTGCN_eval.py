"""`CISTGCN_eval` registry name: the reference ships a byte-identical second copy of the model
file; here it is the same class."""
from .CISTGCN import CISTGCN  # noqa: F401
