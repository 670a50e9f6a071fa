"""Where a tile's lifetime goes (diagnostic build of conv_igemm.hip, -DIG_STAMP): per block wall-clock stamps at start,
first MFMA, end of the K loop, end of block.  python tools/exp/tile_stamps.py C  (1x1 layer, M = 8 x 254 x 254, N = 256)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd import _clx
from cellulus_amd._clx import ClxConvDesc, ClxSrc
_clx.LIB_PATH = os.path.abspath("cellulus_amd/libclx.so.stamp")
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, H, W, N = 8, 254, 254, 256
x = torch.randn(B * H * W, C, device=dev); w = torch.randn(N, C, 1, device=dev) * 0.05; bias = torch.randn(N, device=dev)
wp = torch.empty(N * C, device=dev); st = _clx.stream_ptr(dev)
_clx.call("clx_pack_weights", _clx.ptr(w), _clx.ptr(wp), N, C, 1, C, N, 0, st)
M = B * H * W; out = torch.empty(M, N, device=dev)
d = ClxConvDesc(); d.nsrc = 1
s = ClxSrc(); s.ptr = x.data_ptr(); s.C = C; s.ld = C; s.D, s.H, s.W = 1, H, W; s.oz = s.oy = s.ox = 0; s.fz = s.fy = s.fx = 1
d.src[0] = s; d.B = B; d.ID, d.IH, d.IW = 1, H, W; d.KD, d.KH, d.KW = 1, 1, 1; d.PD = d.PH = d.PW = 0; d.N = N
d.wpack = wp.data_ptr(); d.bias = bias.data_ptr(); d.relu = 1; d.mask = None; d.ld_mask = 0
d.out = out.data_ptr(); d.ld_out = N; d.accumulate = 0; d.algo = 0; d.workspace = None; d.workspace_bytes = 0
for _ in range(3): _clx.call("clx_conv_fwd", ctypes.byref(d), st)
torch.cuda.synchronize()
lib = _clx.load()
nb = ((M + 127) // 128) * 2
buf = (ctypes.c_ulonglong * (8 * nb))()
lib.clx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.clx_debug_stamps(buf, 8 * nb) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
t0 = a[:, 0].min()
start, first, loop_end, end = [(a[:, k] - t0) / 100.0 for k in range(4)]      # microseconds (100 MHz)
print(f"C={C}: {nb} tiles, kernel span {end.max():.1f} us")
print(f"  prologue (start -> first MFMA) median {np.median(first - start):.2f} us  p90 {np.percentile(first - start, 90):.2f}")
print(f"  K loop                         median {np.median(loop_end - first):.2f} us  p90 {np.percentile(loop_end - first, 90):.2f}")
print(f"  epilogue (loop end -> end)     median {np.median(end - loop_end):.2f} us  p90 {np.percentile(end - loop_end, 90):.2f}")
cmax, cmin = (a[:, 7] >> 32) / 100.0, (a[:, 7] & 0xffffffff) / 100.0
nch = C // 32
print(f"     per chunk: mean {np.median((loop_end - first) / nch):.2f} us; a tile's slowest chunk median {np.median(cmax):.2f} p90 {np.percentile(cmax, 90):.2f}; fastest median {np.median(cmin):.2f}")
b1, b2 = (a[:, 5] - t0) / 100.0, (a[:, 6] - t0) / 100.0
print(f"     wave 0 waits for the others   median {np.median(b1 - loop_end):.2f} us  p90 {np.percentile(b1 - loop_end, 90):.2f}")
print(f"     bias + accumulators -> LDS    median {np.median(b2 - b1):.2f} us  p90 {np.percentile(b2 - b1, 90):.2f}")
print(f"     LDS -> global                 median {np.median(end - b2):.2f} us  p90 {np.percentile(end - b2, 90):.2f}")
# slot turnaround: per (xcc, cu, simd-wave-slot ~ hw id) the gap between a block's end and the next block's start on that CU
hw = a[:, 4] & 0xffffffff; xcc = a[:, 4] >> 32
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 0x1) << 7)     # cu_id, se_id, sh_id fields of HW_ID
key = xcc * 1024 + cu
gaps = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    o = idx[np.argsort(start[idx])]
    # two slots per CU: a new block starts when one of the two running ones has ended
    ends = sorted(end[o[:2]].tolist()) if len(o) >= 2 else []
    for j in o[2:]:
        e = ends.pop(0)
        gaps.append(start[j] - e)
        ends.append(end[j]); ends.sort()
gaps = np.array(gaps)
print(f"  CUs seen {len(np.unique(key))}, blocks per CU {nb / len(np.unique(key)):.1f}")
print(f"  slot turnaround (a block's end -> next block's start on that CU) median {np.median(gaps):.2f} us  p90 {np.percentile(gaps, 90):.2f}")
# are the tiles of the whole device in lock step (bursts of prologue loads / epilogue stores)?
w = 4.0
hist = np.bincount((start / w).astype(int))
print("  block starts per %.0f-us window over the kernel:" % w, " ".join(str(v) for v in hist[:60]))
late = start > np.percentile(start, 50)
print(f"  prologue median: first half of the kernel {np.median((first - start)[~late]):.2f} us, second half {np.median((first - start)[late]):.2f} us;"
      f" epilogue {np.median((end - loop_end)[~late]):.2f} / {np.median((end - loop_end)[late]):.2f} us")
