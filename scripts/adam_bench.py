"""micro-benchmark: the fused clip + Adam kernel over a flat buffer of DPOT-M / DPOT-L size (28 B of HBM traffic per parameter)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

for name, n in (("Ti", 7_500_000), ("M", 122_000_000), ("L", 509_000_000)):
    p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4))
    v.abs_()
    hyper = torch.tensor([1e-3, 0.9, 0.9, 1e-8, 1e-6, 0.1, 0.1, 1e9], device="cuda")
    ss = torch.ones(1, device="cuda")
    for _ in range(2): ops.adam_step(p, g, m, v, hyper, ss, 1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): ops.adam_step(p, g, m, v, hyper, ss, 1.0)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 5
    print(f"adam {name}: n = {n/1e6:.1f} M  {t*1e6:8.1f} us  {28.0*n/t/1e12:.2f} TB/s", flush=True)
    del p, g, m, v
