#!/usr/bin/env python
"""launch the bf16x6 GEMM a few times (for rocprofv3 --pmc runs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import ops
for tile in (64, 128):
    for (M, N, K) in ((8192, 512, 2048), (81920, 512, 512)):
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm(A, B, C, M, N, K, transB=True, lda=K, ldb=K, ldc=N, tile=tile, precision=ops.GEMM_BF16X6)
torch.cuda.synchronize()
