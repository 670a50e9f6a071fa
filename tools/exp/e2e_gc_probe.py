"""Is the occasional ~200-ms iteration of bench.train_e2e a generation-2 garbage collection of the main process?
Runs the e2e leg with gc.callbacks timing every collection.    python tools/exp/e2e_gc_probe.py [iterations]"""
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

events, t0 = [], [0.0]


def cb(phase, info):
    if phase == "start":
        t0[0] = time.perf_counter()
    else:
        events.append((info["generation"], (time.perf_counter() - t0[0]) * 1e3, info["collected"]))


gc.callbacks.append(cb)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
res = bench.train_e2e("train2d", dev, iterations=n)
print(json.dumps({k: res[k] for k in ("value", "ms_per_iteration", "iteration_ms_p95", "iteration_ms_max", "loader_wait_ms")}))
big = [(g, round(ms, 1), c) for g, ms, c in events if ms > 2.0]
print("collections:", len(events), "by generation", {g: sum(1 for e in events if e[0] == g) for g in (0, 1, 2)})
print("collections longer than 2 ms (generation, ms, objects collected):", big)
print("gc.get_count()", gc.get_count(), "threshold", gc.get_threshold(), "objects", len(gc.get_objects()))
