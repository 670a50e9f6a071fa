import sys, numpy as np, torch
sys.path.insert(0, '.')
from cellulus_amd.models import get_model
from oracle.unet_oracle import OracleUNetModel
dev = torch.device('cuda:0')
cfg = dict(in_channels=1, out_channels=3, num_fmaps=8, fmap_inc_factor=2, features_in_last_layer=16, downsampling_factors=[[2,2,2]], num_spatial_dims=3)
torch.manual_seed(1)
o32 = OracleUNetModel(**cfg)
for _n, l in o32.named_modules():
    if isinstance(l, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(l.weight, nonlinearity='relu'); torch.nn.init.uniform_(l.bias, -0.1, 0.1)
import copy
o64 = copy.deepcopy(o32).double()
m = get_model(**cfg); m.load_state_dict(o32.state_dict()); m = m.to(dev)
raw = torch.rand(2,1,28,24,32)
r32 = o32(raw); torch.manual_seed(2); dout = torch.randn_like(r32); r32.backward(dout)
r64 = o64(raw.double()); r64.backward(dout.double())
g = m(raw.to(dev)); g.backward(dout.to(dev))
for (n, p32), (_, p64), (_, pm) in zip(o32.named_parameters(), o64.named_parameters(), m.named_parameters()):
    t = p64.grad
    e32 = ((p32.grad.double()-t).norm()/t.norm()).item(); em = ((pm.grad.cpu().double()-t).norm()/t.norm()).item()
    print(f"{n:45s} fp32-oracle {e32:.2e}  hip {em:.2e}")
