#!/usr/bin/env python
"""host->device copy of one DPOT-Tiny batch (xx + yy + msk, B=32) from pinned memory: the PCIe-inclusive figure"""
import time
import torch
B = 32
xs = [torch.randn(B, 128, 128, 10, 4).pin_memory(), torch.randn(B, 128, 128, 1, 4).pin_memory(),
      torch.ones(B, 128, 128, 1, 4).pin_memory()]
ds = [torch.empty_like(x, device="cuda") for x in xs]
nbytes = sum(x.numel() * 4 for x in xs)
for _ in range(3):
    for d, x in zip(ds, xs):
        d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    for d, x in zip(ds, xs):
        d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"batch {nbytes / 1e6:.1f} MB  H2D {dt * 1e3:.3f} ms  {nbytes / dt / 1e9:.1f} GB/s")
