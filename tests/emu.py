"""TEST-ONLY harness: run the shipped HIP kernel sources on the CPU through tests/hipemu so that
kernel logic is exercised by `pytest -m "not gpu"` in the GPU-less build container.

`install()` compiles cistgcn_amd/csrc/*.hip with g++ against the shim and injects the resulting
library into the ctypes loader.  The product package has no reference to this file."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "hipemu"))


def install():
    import build_emu
    from cistgcn_amd import _lib
    path = build_emu.build()
    _lib._handle = _lib.declare(ctypes.CDLL(path))
    _lib._host_pointers_ok = True
    return path


def uninstall():
    from cistgcn_amd import _lib
    _lib._handle = None
    _lib._host_pointers_ok = False
