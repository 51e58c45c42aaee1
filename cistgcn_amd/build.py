"""Builds libcistgcn_hip.so (gfx950) from cistgcn_amd/csrc/*.hip with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.  Every source is
compiled to its own object (in parallel, reused while neither the source nor a header changed) and the objects
are linked into the one shared library the loader opens.
"""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcistgcn_hip.so")
OBJ = os.path.join(os.path.dirname(HERE), "build", "obj")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", CSRC]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers_mtime():
    return max([os.path.getmtime(p) for p in glob.glob(os.path.join(CSRC, "*.h"))] + [os.path.getmtime(__file__)])


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in sources()) or _headers_mtime() > t


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcistgcn_hip.so")
    return hipcc


def build(force=False, verbose=False, extra_flags=(), lib=LIB, obj_dir=OBJ):
    """Compile every HIP source for gfx950 into one shared library; returns its path."""
    if not force and lib == LIB and not _stale():
        return LIB
    hipcc = _hipcc()
    os.makedirs(obj_dir, exist_ok=True)
    hdr = _headers_mtime()

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr):
            return obj, None
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            return obj, "hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr)
        os.replace(obj + ".tmp", obj)
        return obj, None

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, sources()))
    errors = [e for _, e in results if e]
    if errors:
        raise RuntimeError("\n".join(errors))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib + ".tmp"] + [o for o, _ in results]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
