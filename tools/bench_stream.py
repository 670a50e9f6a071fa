"""Streaming (HBM-bound) kernels of the inference path at batch scale: algorithmic bytes /
HIP-event time vs the 8 TB/s HBM3E peak (MI355X_MICROARCH.md).  The table bench.py prints under
infer.streaming (4096^2), here at any size.  Usage: python tools/bench_stream.py [size=8192]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from bench_infer import streaming_rooflines  # noqa: E402
import os
if os.environ.get("CLX_LIB"):
    from cellulus_amd import _clx
    _clx.LIB_PATH = os.path.abspath(os.environ["CLX_LIB"])

if __name__ == "__main__":
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    res = streaming_rooflines(torch.device("cuda:0"), size)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))
    for k, v in res["kernels"].items():
        print(k, json.dumps(v))
