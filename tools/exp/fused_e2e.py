"""Embedding stage of one 512^2 (and 256^2) inference tile with the fused Winograd forms on (default) and off
(CLX_WINO_FUSED=0), each in a process of its own; prints seconds per tile and the per-kind MFMA kernel times."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    from bench_infer import embed_stage
    from cellulus_amd.models import get_model
    dev = torch.device("cuda:0")
    cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
    torch.manual_seed(0)
    model = get_model(**cfg).to(dev)
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    model.eval()
    model.set_infer(p_salt_pepper=0.01, num_infer_iterations=16, device=dev)
    out = {}
    for size in (256, 512):
        t, emb, prof = embed_stage(model, dev, size, 16, 3)
        out[size] = dict(ms=round(t * 1e3, 2), mpix_per_s=round(size * size / t / 1e6, 3),
                         kinds={k: [round(v[0]), round(v[1], 2), round(v[2] / 1e12, 3)] for k, v in prof.items() if v[0]},
                         checksum=float(emb.double().abs().sum()))
        plan = next(iter(model._plans.values()))
        out[size]["algo_fwd"] = {n: a["fwd"] for n, a in plan.algo.items() if a["fwd"]}
        out[size]["subpixel"] = {n: [sp["wino"], sp["wino_skip"], sp["fused_z"], sp["fused_skip"]] for n, sp in plan.subpixel.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    for label, env in (("fused", {}), ("three_launch", {"CLX_WINO_FUSED": "0"}), ("fused_all", {"CLX_WINO_FUSED_MAX_CHANNELS": "4096"})):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(label, line[-1] if line else r.stderr[-2000:], flush=True)
