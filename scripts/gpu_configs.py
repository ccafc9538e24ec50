"""Sanity + timing of the BASELINE.json configs beyond Tiny (Small / Medium / Large): forward parity against the CPU
oracle at B=1 (the oracle is the checker only) and the time of one eager + one graphed train step at a larger batch.
Writes one JSON line per config.  Usage: python scripts/gpu_configs.py [S M L]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dpot_ref as R                                       # noqa: E402  (checker)
from dpot_amd import DPOTNet                                           # noqa: E402
from dpot_amd.train import FlatParams, FusedAdam, GraphedTrainStep     # noqa: E402

CFGS = {"S": (R.SMALL, 16), "M": (R.MEDIUM, 16), "L": (R.LARGE, 4), "T": (R.TINY, 32)}


def main():
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    for key in (sys.argv[1:] or ["S", "M", "L"]):
        kw, Bt = CFGS[key]
        cfg = R.DPOTConfig(**kw)
        sd = R.recipe_state_dict(cfg)
        model = DPOTNet(**kw).cuda()
        model.load_state_dict(sd)
        x = R.recipe_input((1, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels))
        t0 = time.time()
        with torch.no_grad():
            ref, ref_cls = R.dpot_forward(sd, x, cfg)
        t_cpu = time.time() - t0
        with torch.no_grad():
            out, cls = model(x.cuda())
        err = (out.cpu().double() - ref.double()).abs().max().item() / ref.double().abs().max().item()
        errc = (cls.cpu().double() - ref_cls.double()).abs().max().item() / ref_cls.double().abs().max().item()
        # train step timing
        S = cfg.img_size
        xx = torch.randn(Bt, S, S, cfg.in_timesteps, cfg.in_channels, device="cuda")
        yy = torch.randn(Bt, S, S, 1, cfg.out_channels, device="cuda")
        msk = torch.ones(Bt, S, S, 1, cfg.out_channels, device="cuda")
        opt = FusedAdam(FlatParams(model), lr=1e-4, weight_decay=1e-6, max_norm=10000.0)
        g = GraphedTrainStep(model, opt, xx, yy, msk)
        for _ in range(3):
            loss = g.replay(lr=1e-4)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 10
        for _ in range(n):
            loss = g.replay(lr=1e-4)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / n * 1e3
        print(json.dumps({"config": key, "params_M": round(sum(p.numel() for p in model.parameters()) / 1e6, 1),
                          "fwd_rel_err_vs_oracle": err, "cls_rel_err": errc, "oracle_fwd_s_B1": round(t_cpu, 2),
                          "train_batch": Bt, "train_ms_per_step": round(ms, 3),
                          "samples_per_s": round(Bt / ms * 1e3, 1), "loss": float(loss),
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)
        del model, opt, g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
