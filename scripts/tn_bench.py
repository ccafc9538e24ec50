"""weight-gradient GEMM micro-benchmark: csrc/gemm_tn.hip vs the generic kernel (DPOT_GEMM_TN=0), graph-timed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for (M, N, K, batch, sks) in ((512, 512, 8192, 1, (4, 8, 16, 32)), (256, 256, 4608, 4, (4, 8, 16)), (512, 2048, 8192, 1, (2, 4, 8))):
    A = torch.randn(K, M * batch, device="cuda"); B = torch.randn(K, N * batch, device="cuda")
    C = torch.empty(batch, M, N, device="cuda"); cs = torch.empty(batch, M, device="cuda")
    fl = 2.0 * M * N * K * batch
    for sk in sks:
        t = timeit(lambda: ops.gemm(A, B, C, M, N, K, transA=True, lda=M * batch, ldb=N * batch, ldc=N, batch=batch, strideA=M,
                                    strideB=N, strideC=M * N, splitk=sk, colsum_out=cs, colsum_of=1, strideColsum=M))
        print(f"M={M} N={N} K={K} batch={batch} splitk={sk}: {t:7.1f} us  {fl/t/1e6:6.1f} TF (incl. reduce)", flush=True)
