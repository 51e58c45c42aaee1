"""TEST-ONLY: compile the shipped .hip sources with g++ against the HIP shim in this directory,
producing tests/hipemu/_build/libcistgcn_emu.so (git-ignored; never shipped, never loaded by the
product package)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "cistgcn_amd", "csrc")
# HIPEMU_SANITIZE=1: AddressSanitizer build of the very same sources (run python under LD_PRELOAD=$(g++ -print-file-name=libasan.so)
# with ASAN_OPTIONS=detect_leaks=0); the GPU pool offers no device sanitizer, this is where out-of-bounds indexing is hunted.
SANITIZE = os.environ.get("HIPEMU_SANITIZE", "0") == "1"
OUT = os.path.join(HERE, "_build", "libcistgcn_emu_asan.so" if SANITIZE else "libcistgcn_emu.so")


def build(force=False):
    """Idempotent and safe under concurrent callers (pytest workers, spawned ranks): an exclusive file lock serialises the
    builders and the library appears under its final name only when complete."""
    import fcntl
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force):
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    deps = srcs + hdrs + [os.path.join(HERE, "emu_runtime.cpp")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(p) <= os.path.getmtime(OUT) for p in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hdr_time = max(os.path.getmtime(p) for p in hdrs)

    def compile_one(s):
        # one object per source, reused while neither the source nor a header changed
        o = os.path.join(HERE, "_build", os.path.basename(s) + (".asan.o" if SANITIZE else ".o"))
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), hdr_time):
            return o, None
        cmd = ["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-pthread", "-Wno-unknown-pragmas", "-Wno-attributes",
               "-I", HERE, "-I", CSRC, "-c", s, "-o", o + ".tmp"] + (["-fsanitize=address", "-fno-omit-frame-pointer"] if SANITIZE else [])
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            return o, "g++ failed on %s:\n%s" % (s, res.stderr[-4000:])
        os.replace(o + ".tmp", o)
        return o, None

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, srcs + [os.path.join(HERE, "emu_runtime.cpp")]))
    errors = [e for _, e in results if e]
    if errors:
        raise RuntimeError("\n".join(errors))
    objs = [o for o, _ in results]
    tmp = OUT + ".tmp.%d" % os.getpid()
    res = subprocess.run(["g++", "-shared", "-pthread", "-o", tmp] + objs + (["-fsanitize=address"] if SANITIZE else []), capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stderr[-4000:])
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
