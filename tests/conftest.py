import os
import sys

import pytest

# The plane / tile kernel generations of the fused ST-GCN stage are pinned by some tests through cg_stgcn_domain_planes_min_workgroups; the
# library honours that switch only in a process started with CISTGCN_ABLATION=1 (a production process cannot have it changed under it).
os.environ.setdefault("CISTGCN_ABLATION", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
