"""Experiment: one batch of 8 on one stream against two half batches on two streams
(HBM-bound transform kernels of one half under the MFMA-bound GEMMs of the other)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cellulus_amd.models import get_model
from cellulus_amd.models.plan import UNetPlan, build_topology

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "train2d"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = get_model(**wl["model"]).to(dev)
for _n, layer in model.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
model.flatten_parameters()
params = model._ordered_params()
grads = model.attach_flat_grads()
B = wl["batch"]
raw = bench.synthetic_raw(B, wl["crop"], seed=0).to(dev)

def make(b):
    topo = build_topology(model.in_channels, model.out_channels, model.num_fmaps, model.fmap_inc_factor,
                          model.features_in_last_layer, model.downsampling_factors, model.num_spatial_dims,
                          raw.shape[2:])
    p = UNetPlan(topo, b, dev, True)
    p.pack_weights(params, 1, need_dgrad=True)
    return p

JOIN = int(os.environ.get("JOIN", "0"))     # 1: join at the end of a step, 2: also between forward and backward

def join(streams):
    main = torch.cuda.current_stream()
    for s in set(streams):
        main.wait_stream(s)
    for s in set(streams):
        s.wait_stream(main)

DELAY_US = float(os.environ.get("DELAY_US", "0"))      # the second stream starts this much later (device-side spin)

def run(plans, streams, raws, steps):
    outs = []
    for k, (p, s, r) in enumerate(zip(plans, streams, raws)):
        with torch.cuda.stream(s):
            if k == 1 and DELAY_US > 0 and len(set(streams)) > 1:
                torch.cuda._sleep(int(DELAY_US * 2100))
            outs.append(p.forward(r, params))
    if JOIN >= 2:
        join(streams)
    for p, s, o in zip(plans, streams, outs):
        with torch.cuda.stream(s):
            p.backward(torch.ones_like(o), params, grads, flat_grad=model._flat_grad)
    if JOIN >= 1:
        join(streams)

def timeit(plans, streams, raws, steps=6):
    for _ in range(2):
        run(plans, streams, raws, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(plans, streams, raws, 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

full = make(B)
print("one stream, batch %d: %.2f ms" % (B, timeit([full], [torch.cuda.current_stream()], [raw])))
del full
torch.cuda.empty_cache()
h = B // 2
pa, pb = make(h), make(h)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
print("one stream, 2 x batch %d back to back: %.2f ms" % (h, timeit([pa, pb], [s0, s0], [raw[:h].contiguous(), raw[h:].contiguous()])))
print("two streams, 2 x batch %d: %.2f ms" % (h, timeit([pa, pb], [s0, s1], [raw[:h].contiguous(), raw[h:].contiguous()])))
if DELAY_US > 0:
    sys.exit(0)
del pa, pb
torch.cuda.empty_cache()
pa, pb = make(h), make(h)
lo, hi = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)
print("two streams (one high priority), 2 x batch %d: %.2f ms" % (h, timeit([pa, pb], [lo, hi], [raw[:h].contiguous(), raw[h:].contiguous()])))
del pa, pb
torch.cuda.empty_cache()
q = B // 4
ps = [make(q) for _ in range(4)]
ss = [torch.cuda.Stream() for _ in range(4)]
rs = [raw[i * q:(i + 1) * q].contiguous() for i in range(4)]
print("four streams, 4 x batch %d: %.2f ms" % (q, timeit(ps, ss, rs)))
print("two streams, 4 x batch %d: %.2f ms" % (q, timeit(ps, [ss[0], ss[1], ss[0], ss[1]], rs)))
