"""diagnostic: per-tensor gradient error of the HIP path and of the fp32 CPU oracle, both against the fp64 oracle"""
import sys, os
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dpot_ref as R

name, B = sys.argv[1], int(sys.argv[2])
kw = getattr(R, name)
cfg = R.DPOTConfig(**kw)
S = cfg.img_size
x = R.recipe_input((B, S, S, cfg.in_timesteps, cfg.in_channels), salt=71)
up_y = R.recipe_input((B, S, S, cfg.out_timesteps, cfg.out_channels), salt=72) * 0.3
up_c = R.recipe_input((B, cfg.n_cls), salt=73) * 0.3


def oracle(dt):
    sd = OrderedDict((k, v.clone().to(dt).requires_grad_(True)) for k, v in R.recipe_state_dict(cfg, salt=4).items())
    xo = x.clone().to(dt).requires_grad_(True)
    yo, co = R.dpot_forward(sd, xo, cfg)
    ((yo * up_y.to(dt)).sum() + (co * up_c.to(dt)).sum()).backward()
    return {k: v.grad for k, v in sd.items()}, xo.grad


g64, dx64 = oracle(torch.float64)
g32, dx32 = oracle(torch.float32)
res = {}
if torch.cuda.is_available():
    from dpot_amd import DPOTNet
    m = DPOTNet(**kw)
    m.load_state_dict(R.recipe_state_dict(cfg, salt=4))
    m.cuda()
    xg = x.cuda().requires_grad_(True)
    y, c = m(xg)
    ((y * up_y.cuda()).sum() + (c * up_c.cuda()).sum()).backward()
    res = {k: p.grad.cpu() for k, p in m.named_parameters()}
print(f"{'tensor':44s} {'|g64|':>11s} {'cpu32 relnorm':>14s} {'hip relnorm':>12s} {'cpu32 |n| err':>14s} {'hip |n| err':>12s}")
for k, g in g64.items():
    n = g.norm().item()
    e32 = (g32[k].double() - g).norm().item() / (n + 1e-300)
    ne32 = abs(g32[k].double().norm().item() - n) / (n + 1e-300)
    if k in res:
        eh = (res[k].double() - g).norm().item() / (n + 1e-300)
        neh = abs(res[k].double().norm().item() - n) / (n + 1e-300)
    else:
        eh = neh = float("nan")
    flag = " <<<" if max(ne32, neh) > 5e-5 else ""
    print(f"{k:44s} {n:11.4e} {e32:14.2e} {eh:12.2e} {ne32:14.2e} {neh:12.2e}{flag}")
