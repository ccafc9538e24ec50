#!/usr/bin/env python
"""time the bf16x6 kernel on a few shapes (kernel experiments; pick the library with DPOT_HIP_LIB)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import ops  # noqa: E402


def run(name, M, N, K, transA=False, transB=True, batch=1, splitk=1, tile=128, reps=30):
    dev = "cuda"
    A = torch.randn(*((batch, K, M) if transA else (batch, M, K)), device=dev)
    B = torch.randn(*((batch, N, K) if transB else (batch, K, N)), device=dev)
    C = torch.empty(batch, M, N, device=dev)
    kw = dict(transA=transA, transB=transB, lda=A.shape[2], ldb=B.shape[2], ldc=N, batch=batch,
              strideA=A.shape[1] * A.shape[2], strideB=B.shape[1] * B.shape[2], strideC=M * N, splitk=splitk)
    for _ in range(3):
        ops.gemm(A, B, C, M, N, K, tile=tile, precision=ops.GEMM_BF16X6, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(A, B, C, M, N, K, tile=tile, precision=ops.GEMM_BF16X6, **kw)
    e1.record()
    e1.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    fl = 2.0 * M * N * K * batch
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:6d} tile={tile:3d} {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF", flush=True)


if __name__ == "__main__":
    print(os.environ.get("DPOT_HIP_LIB", "default"))
    for tile in (128, 64):
        run("NT big", 81920, 512, 512, tile=tile)
        run("NT MLP", 8192, 512, 512, tile=tile)
        run("NN out", 8192, 2048, 512, transB=False, tile=tile)
        run("NT K2048", 8192, 512, 2048, tile=tile)
        run("TN wgrad sk4", 512, 2048, 8192, transA=True, transB=False, splitk=4, tile=tile)
