#!/usr/bin/env python
"""launch a few big GEMMs (for rocprofv3 --pmc runs)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import ops
for tile in (64, 128):
    M, N, K = 8192, 512, 5120
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.empty(M, N, device="cuda")
    for _ in range(6):
        ops.gemm(A, B, C, M, N, K, transB=True, lda=K, ldb=K, ldc=N, tile=tile)
torch.cuda.synchronize()
