"""Is the chain-vs-plain gradient difference under CLX_PRECISION=f32x3bf16 a flipped ReLU gate (seed-dependent) or systematic?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import test_gpu_unet as T
dev = torch.device("cuda:0")
for seed in (4, 5, 6, 7, 8):
    os.environ.pop("CLX_CHAIN64", None)
    _o, model, raw = T._make("2d_chain64", dev, seed=seed)
    x = raw.to(dev)
    got = model(x)
    torch.manual_seed(9)
    dout = torch.randn_like(got)
    got.backward(dout)
    grads = [p.grad.clone() for p in model.parameters()]
    os.environ["CLX_CHAIN64"] = "0"
    _o, plain, _r = T._make("2d_chain64", dev, seed=seed)
    ref = plain(x)
    ref.backward(dout)
    worst = max((((p.grad - g).norm() / (p.grad.norm() + 1e-30)).item(), n) for (n, p), g in zip(plain.named_parameters(), grads))
    print(f"seed {seed}: max |out diff| {(ref - got).abs().max().item():.2e}  worst grad rel-L2 {worst[0]:.2e} ({worst[1]})", flush=True)
