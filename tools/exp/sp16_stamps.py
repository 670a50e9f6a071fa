"""Prints the s_memtime stamps of the diagnostic build of gemm_sp16_kernel (tools/build_variant.sh stamp gemm_sp.hip -DSP16_STAMP):
cycles between the segments of double steps 16 .. 19 for an early wave (0) and a late wave (4) of block 0.
    CLX_LIB=cellulus_amd/libclx.so.stamp python tools/exp/sp16_stamps.py [M N K]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd import _clx
os.environ["CLX_SP_MFMA"] = "16"
_clx.LIB_PATH = os.path.abspath(os.environ["CLX_LIB"])
lib = _clx.load()
dev = torch.device("cuda:0")
st = _clx.stream_ptr(dev)
M, N, K = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (123008, 768, 2304)
a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
def planes(x):
    buf = torch.empty(lib.clx_planes_bytes(x.shape[0], x.shape[1]), dtype=torch.uint8, device=dev)
    _clx.call("clx_split_planes", _clx.ptr(x), x.stride(0), x.shape[0], x.shape[1], _clx.ptr(buf), st)
    return buf
pa, pb = planes(a), planes(b)
out = torch.empty(M, N, device=dev)
lib.clx_gemm_planes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
for _ in range(20):
    lib.clx_gemm_planes(_clx.ptr(pa), _clx.ptr(pb), M, N, K, None, 0, _clx.ptr(out), N, st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib.clx_sp16_stamps.argtypes = [ctypes.c_void_p]
assert lib.clx_sp16_stamps(buf) == 0
names = {0: ["at OPEN", "through OPEN", "reads issued", "B requested, fragments here", "at MID (H0 products issued)", "through MID", "reads issued", "A requested, B23 here"],
         1: ["at OPEN", "through OPEN", "H1(prev) issued", "reads issued", "at MID (B requested; A, B01 here)", "through MID", "H0 issued", "reads issued"]}
for kind in (0, 1):
    t = [buf[kind * 64 + i] for i in range(64)]
    t = [v for v in t if v]
    print("early wave 0" if kind == 0 else "late wave 4", len(t), "stamps")
    for i in range(1, len(t)):
        print(f"  d={16 + i // 8} {names[kind][i % 8]:30s} +{t[i] - t[i - 1]:6d}   (since first {t[i] - t[0]})")
