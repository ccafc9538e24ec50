"""eager train step of a BASELINE config under rocprofv3 (kernel census): python scripts/census_config.py M bf16 [auto]
(third argument: gemm_precision of the model - bench.py runs S / M / L with "auto")"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import DPOTNet, ops
from dpot_amd.train import FlatParams, FusedAdam, rollout
import bench

CFGS = {k: (v[1], v[2], 1) for k, v in bench.CONFIGS.items() if k in ("T", "S", "M", "L")}

key, mlp = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
kw, B, T_ar = CFGS[key]
B = int(os.environ.get("CENSUS_BATCH", B))          # CENSUS_BATCH=16: DPOT-L at the batch bench.py measures it
ops.set_mlp_precision(mlp)
model = DPOTNet(**kw).cuda()
if len(sys.argv) > 3:
    model.gemm_precision = sys.argv[3]
S = kw["img_size"]
xx = torch.randn(B, S, S, 10, 4, device="cuda"); yy = torch.randn(B, S, S, T_ar, 4, device="cuda"); msk = torch.ones(B, S, S, 1, 4, device="cuda")
opt = FusedAdam(FlatParams(model), lr=1e-4, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=10000.0)
for _ in range(4):
    opt.zero_grad()
    loss, _ = rollout(model, xx, yy, msk, noise_scale=0.0005)
    loss.backward()
    opt.step(lr=1e-4)
torch.cuda.synchronize()
print("ok", float(loss))
