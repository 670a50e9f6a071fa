"""A/B of the connected-component passes at 4096^2 (run one process per variant: the switches are read once).
    CLX_CC_MASKED_REWRITE=0|1  CLX_CC_LABELS=0|1  python tools/exp/cc_variants.py
Prints the kernels' time per call (libclx's event pairs, kind CLX_PROF_CC; per kernel: run it under rocprofv3 --kernel-trace --stats) and checks the labels against the
pixel-list path's (computed in a child process with CLX_CC_LABELS=0)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from cellulus_amd import _clx
from cellulus_amd.utils.misc import label_on_device


def image(n, seed=0):
    """blobs of distinct values on a grid, as the streaming table's label map (about a quarter foreground)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n]
    cy, cx = (yy // 48) * 48 + 24, (xx // 48) * 48 + 24
    r = 8 + ((cy * 7 + cx * 13) % 11)
    inside = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    val = ((cy // 48) * (n // 48 + 1) + cx // 48 + 1) % 60000 + 1
    seg = np.where(inside, val, 0).astype(np.int32)
    seg[rng.random((n, n)) < float(os.environ.get("CC_SPECKS", "0.0002"))] = 5            # specks the size filter removes
    return seg


def main():
    n = int(os.environ.get("CC_N", "4096"))
    dev = torch.device("cuda:0")
    seg = torch.from_numpy(image(n)).to(dev)
    out, ncomp = label_on_device(seg, 70)
    torch.cuda.synchronize()
    if len(sys.argv) > 1 and sys.argv[1] == "--dump":
        np.save(sys.argv[2], out.cpu().numpy())
        return
    from bench_infer import PROF_KIND, _kernel_time
    t_call, t_k, n_k = min((_kernel_time(lambda: label_on_device(seg, 70), PROF_KIND["cc"], reps=5) for _ in range(3)),
                           key=lambda r: r[1])
    print({k: os.environ.get(k) for k in ("CLX_CC_LABELS", "CLX_CC_MASKED_REWRITE")},
          f"kernels {t_k * 1e6:.1f} us in {n_k:.0f} launches, call {t_call * 1e6:.1f} us,",
          "frac of 8 TB/s", round(n * n * 8 / t_k / 8e12, 3), "ncomp", int(ncomp.item()))
    ref_path = "/tmp/cc_ref_%d.npy" % n
    if not os.path.exists(ref_path):
        # (never from a profiled process: the child would inherit the profiler's preload, start from a process that has
        #  initialised the GPU, and its CLX_CC_LABELS=0 kernels would land in the same counter files — cc_pmc.sh writes
        #  the reference first, outside the profiler)
        if any(k.startswith("ROCPROF") or k == "ROCP_TOOL_LIBRARIES" for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
            print("   (no reference dump and running under the profiler: comparison skipped)")
            return
        env = dict(os.environ, CLX_CC_LABELS="0")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--dump", ref_path], env=env, check=True)
    ref = np.load(ref_path)
    print("   identical to the pixel-list path:", bool((ref == out.cpu().numpy()).all()))


if __name__ == "__main__":
    main()
