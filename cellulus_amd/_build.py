"""Builds libclx.so (the HIP kernel library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container and
the resulting ``cellulus_amd/libclx.so`` travels to the GPU box with the
snapshot.  Only out-of-date objects are recompiled.
"""

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libclx.so")
ARCH = "gfx950"

COMMON_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
]
# float64 clustering/labelling code must not fuse multiply-adds: the membership
# test d^2 <= bw^2 is compared with the reference's un-fused arithmetic.
PER_FILE_FLAGS = {
    "meanshift.hip": ["-ffp-contract=off"],
    "seeds.hip": ["-ffp-contract=off"],
    # the fused Winograd kernels spell their fused multiply-adds out: a tile gets the same bits wherever it sits in a block
    "wino_fused.hip": ["-ffp-contract=off"],
}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libclx.so")
    return exe


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "clx.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile_one(args):
    hipcc, src, obj, flags = args
    extra = os.environ.get("CLX_EXTRA_HIPCC_FLAGS", "").split()
    cmd = [hipcc, *COMMON_FLAGS, *flags, *extra, "-c", src, "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
    return obj


def build(force=False, verbose=False):
    """Compile every .hip under csrc/ and link cellulus_amd/libclx.so."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_mtime = _newest_header_mtime()
    jobs, objs = [], []
    for name in sources():
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ_DIR, name[:-4] + ".o")
        objs.append(obj)
        stale = (
            force
            or not os.path.exists(obj)
            or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_mtime)
        )
        if stale:
            jobs.append((hipcc, src, obj, PER_FILE_FLAGS.get(name, [])))
    if jobs:
        if verbose:
            print(f"[cellulus_amd] compiling {len(jobs)} HIP sources for {ARCH}", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as pool:
            list(pool.map(_compile_one, jobs))
    need_link = bool(jobs) or not os.path.exists(LIB_PATH)
    if not need_link:
        need_link = os.path.getmtime(LIB_PATH) < max(os.path.getmtime(o) for o in objs)
    if need_link:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
