"""Builds libcistgcn_hip.so (gfx950) from cistgcn_amd/csrc/*.hip with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcistgcn_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared library; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcistgcn_hip.so")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-I", CSRC, "-o", LIB + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
