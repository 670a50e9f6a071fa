"""Layer-level timing of the one-launch Winograd forward (CLX_ALGO_WINOGRAD4_FUSED, csrc/wino_fused.hip) against the
three-launch form (CLX_ALGO_WINOGRAD4), in float32 MFMA and in the split precision, on the Winograd layers of the benchmark configurations: the inference chunk
(8 noisy copies of a 512^2 tile, cfg-5) and the training half batch (4 crops of 256^2, cfg-2).

    python tools/exp/fused_bench.py [--only NAME] [--reps 5] > gpurun_out/fused_bench.txt
"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd import _clx  # noqa: E402
from cellulus_amd._clx import ClxConvDesc, ClxSrc  # noqa: E402

# name, B, H (= W) of the input, C, N, kernel, accumulate
LAYERS = [
    ("infer l0.6 256->256", 8, 526, 256, 256, 3, 0),
    ("infer l1.0 256->768", 8, 262, 256, 768, 3, 0),
    ("infer l1.6 768->768", 8, 260, 768, 768, 3, 0),
    ("infer r0.0 skip 256->64", 8, 516, 256, 64, 3, 1),
    ("infer r0.0 low 768->256 (2x2)", 8, 258, 768, 256, 2, 0),
    ("infer r0.6 64->64", 8, 514, 64, 64, 3, 0),
    ("train l0.6 256->256", 4, 254, 256, 256, 3, 0),
    ("train l1.0 256->768", 4, 126, 256, 768, 3, 0),
    ("train l1.6 768->768", 4, 124, 768, 768, 3, 0),
    ("train r0.0 skip 256->64", 4, 244, 256, 64, 3, 1),
    ("train r0.0 low 768->256 (2x2)", 4, 122, 768, 256, 2, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _clx.load()
    st = _clx.stream_ptr(dev)
    for name, B, H, C, N, k, acc in LAYERS:
        if args.only and args.only not in name:
            continue
        OH = H - k + 1
        x = torch.randn(B, H, H, C, device=dev)
        w = (torch.randn(N, C, k * k, device=dev) * 0.05).contiguous()
        bias = torch.randn(N, device=dev)
        nxi = (4 + k - 1) ** 2
        packs = {}
        for mode in (4, 7):
            packs[mode] = torch.empty(nxi * N * C, device=dev)
            _clx.call("clx_pack_weights", _clx.ptr(w), _clx.ptr(packs[mode]), N, C, k * k, C, N, mode, st)
        out = {a: torch.zeros(B, OH, OH, N, device=dev) for a in (2, 3, "sp")}
        # the three-launch form in the split precision: planes of the packed weights
        sp_ok = N % 128 == 0 and C % 128 == 0
        if sp_ok:
            wplanes = torch.empty(int(lib.clx_planes_bytes(nxi * N, C)), dtype=torch.uint8, device=dev)
            _clx.call("clx_split_planes", _clx.ptr(packs[4]), C, nxi * N, C, _clx.ptr(wplanes), st)

        def desc(algo, sp=False):
            d = ClxConvDesc()
            d.nsrc = 1
            s = ClxSrc()
            s.ptr, s.C, s.ld = x.data_ptr(), C, C
            s.D, s.H, s.W = 1, H, H
            s.fz = s.fy = s.fx = 1
            d.src[0] = s
            d.B = B
            d.ID, d.IH, d.IW = 1, H, H
            d.KD, d.KH, d.KW = 1, k, k
            d.N = N
            d.algo = algo
            d.bias, d.relu, d.accumulate = bias.data_ptr(), 1, acc
            d.out, d.ld_out = out["sp" if sp else algo].data_ptr(), N
            d.wpack = packs[4 if algo == 2 else 7].data_ptr()
            if sp:
                d.precision = 1
                d.wplanes = wplanes.data_ptr()
            return d

        d2 = desc(2)
        need = int(lib.clx_conv_workspace_bytes(ctypes.byref(d2), 0))
        ws = torch.empty(need // 4 + 4, device=dev)
        d2.workspace, d2.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        d3 = desc(3)
        need3 = int(lib.clx_conv_fused_workspace_bytes(ctypes.byref(d3)))
        assert need3 <= ws.numel() * 4
        if need3 and os.environ.get("CLX_FUSED_NO_WS") is None:
            d3.workspace, d3.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        tiles = B * (-(-OH // 4)) ** 2
        flops = 2.0 * nxi * tiles * N * C
        res = dict(layer=name, tiles=tiles, gflop_executed=flops / 1e9)
        runs = [("three_launch", d2), ("fused", d3)]
        if sp_ok:
            dsp = desc(2, sp=True)
            need_sp = int(lib.clx_conv_workspace_bytes(ctypes.byref(dsp), 0))
            ws_sp = torch.empty(need_sp // 4 + 4, device=dev)
            dsp.workspace, dsp.workspace_bytes = ws_sp.data_ptr(), ws_sp.numel() * 4
            runs.append(("three_launch_sp", dsp))
        for key, d in runs:
            for _ in range(2):
                _clx.call("clx_conv_fwd", ctypes.byref(d), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                _clx.call("clx_conv_fwd", ctypes.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            res[key + "_ms"] = round(ms, 4)
            res[key + "_tflops"] = round(flops / ms / 1e9, 1)
        if not acc:
            res["max_abs_diff"] = float((out[2] - out[3]).abs().max())
        res["speedup"] = round(res["three_launch_ms"] / res["fused_ms"], 3)
        if sp_ok:
            res["fused_over_sp"] = round(res["fused_ms"] / res["three_launch_sp_ms"], 3)
            if not acc:
                res["max_abs_diff_sp"] = float((out[2] - out["sp"]).abs().max())
        print(json.dumps(res), flush=True)
        del x, w, ws, out, packs


if __name__ == "__main__":
    main()
