"""Timing of the BASELINE.json configs (Tiny / Small / Medium / Large) in the GEMM precision modes, plus configs[4]:
the 20-step auto-regressive DPOT-Large rollout train step with activation recomputation (peak memory reported).
One JSON line per (config, mode).  Usage: python scripts/gpu_configs2.py [T S M L L20]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import DPOTNet, ops                                       # noqa: E402
from dpot_amd.train import FlatParams, FusedAdam, GraphedTrainStep     # noqa: E402

TINY = dict(img_size=128, patch_size=8, in_channels=4, out_channels=4, in_timesteps=10, out_timesteps=1, n_blocks=4,
            embed_dim=512, out_layer_dim=32, depth=4, modes=32, mlp_ratio=1, n_cls=12)
CFGS = {"T": (TINY, 32, 1), "S": (dict(TINY, embed_dim=1024, depth=6, n_blocks=8), 32, 1),
        "M": (dict(TINY, embed_dim=1024, depth=12, n_blocks=8, mlp_ratio=4), 32, 1),
        "L": (dict(TINY, img_size=256, embed_dim=1536, depth=24, n_blocks=16, mlp_ratio=4, out_layer_dim=128, modes=64), 4, 1),
        "L20": (dict(TINY, img_size=256, embed_dim=1536, depth=24, n_blocks=16, mlp_ratio=4, out_layer_dim=128, modes=64), 4, 20)}
MODES = [("f32", None), ("auto", None), ("f32", "bf16x6"), ("f32", "bf16")]      # (gemm precision, channel-MLP override)


def main():
    for key in (sys.argv[1:] or ["T", "S", "M", "L", "L20"]):
        kw, B, T_ar = CFGS[key]
        for gp, mp in (MODES if T_ar == 1 else [MODES[0], MODES[1], MODES[3]]):
            torch.manual_seed(0)
            ops.set_gemm_precision(gp)
            ops.set_mlp_precision(mp)
            model = DPOTNet(**kw).cuda()
            model.recompute_blocks = T_ar > 1
            S = kw["img_size"]
            xx = torch.randn(B, S, S, 10, 4, device="cuda")
            yy = torch.randn(B, S, S, T_ar, 4, device="cuda")
            msk = torch.ones(B, S, S, 1, 4, device="cuda")
            opt = FusedAdam(FlatParams(model), lr=1e-4, betas=(0.9, 0.9), weight_decay=1e-6, max_norm=10000.0)
            torch.cuda.reset_peak_memory_stats()
            g = GraphedTrainStep(model, opt, xx, yy, msk, noise_scale=0.0005, warmup=1)
            for _ in range(2):
                loss = g.replay(lr=1e-4)
            torch.cuda.synchronize()
            n = 10 if T_ar == 1 else 3
            t0 = time.time()
            for _ in range(n):
                loss = g.replay(lr=1e-4)
            torch.cuda.synchronize()
            ms = (time.time() - t0) / n * 1e3
            print(json.dumps({"config": key, "gemm": gp, "mlp": mp or gp, "batch": B, "T_ar": T_ar,
                              "recompute": T_ar > 1, "ms_per_step": round(ms, 3),
                              "sample_steps_per_s": round(B * T_ar / ms * 1e3, 1), "loss": round(float(loss), 4),
                              "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
            del model, opt, g
            torch.cuda.empty_cache()
    ops.set_gemm_precision("f32")
    ops.set_mlp_precision(None)


if __name__ == "__main__":
    main()
