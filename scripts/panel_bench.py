"""micro-benchmark of the panel GEMM (csrc/gemm_panel.hip) vs the generic kernel; DPOT_PANEL_RT=1..5 forces the panel height"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps

if __name__ == "__main__":
  for M, N, K in ((8192, 512, 512), (8192, 2048, 512), (8192, 512, 2048), (8192, 4096, 1024), (4096, 6144, 1536)):
      A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
      pk = ops.PanelPacks([(W, N, K, K, False)]); pk.refresh()
      t_p = timeit(lambda: ops.gemm_panel(A, pk.bufs[0], N, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True))
      t_g = timeit(lambda: ops.linear_fwd(A, W, b, act=1, save_pre=True))
      t_pl = timeit(lambda: ops.gemm_panel(A, pk.bufs[0], N))
      t_gl = timeit(lambda: ops.linear_fwd(A, W, None))
      fl = 2.0 * M * N * K
      print(f"RT={os.environ.get('DPOT_PANEL_RT','auto')} M={M} N={N} K={K}: panel gelu+pre {t_p*1e6:7.1f} us {fl/t_p/1e12:6.1f} TF | generic {t_g*1e6:7.1f} us {fl/t_g/1e12:6.1f} TF"
            f" || linear: panel {t_pl*1e6:7.1f} us {fl/t_pl/1e12:6.1f} TF | generic {t_gl*1e6:7.1f} us {fl/t_gl/1e12:6.1f} TF", flush=True)
