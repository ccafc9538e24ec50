"""the fp32 panel GEMM at the DPOT-L out-layer shape (tokens x 2048 x 1536) and neighbours: where does it fall off?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops
from scripts.panel_bench import timeit

shapes = [(16384, 2048, 1536), (16384, 2048, 1024), (16384, 2048, 1280), (16384, 2048, 2048), (16384, 2048, 512),
          (16384, 1024, 1536), (16384, 512, 1536), (8192, 2048, 1536), (4096, 2048, 1536), (16384, 1536, 2048)]
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    pk = ops.PanelPacks([(W, N, K, K, False)]); pk.refresh()
    t_pl = timeit(lambda: ops.gemm_panel(A, pk.bufs[0], N, bias=b), reps=10)
    t_gl = timeit(lambda: ops.linear_fwd(A, W, b), reps=10)
    fl = 2.0 * M * N * K
    print(f"RT={os.environ.get('DPOT_PANEL_RT','auto')} M={M} N={N} K={K}: panel {t_pl*1e6:8.1f} us {fl/t_pl/1e12:6.1f} TF | generic {t_gl*1e6:8.1f} us {fl/t_gl/1e12:6.1f} TF", flush=True)
    del A, W, pk
