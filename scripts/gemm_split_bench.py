#!/usr/bin/env python
"""native fp32 MFMA vs bf16x6 split GEMM: accuracy against an fp64 product and time per launch, DPOT shapes"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import ops  # noqa: E402


def run(name, M, N, K, transA=False, transB=True, batch=1, splitk=1, tiles=(64, 128), reps=30, scale=1.0):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(M + 7 * N + 13 * K)
    A = torch.randn(*((batch, K, M) if transA else (batch, M, K)), device=dev, generator=g) * scale
    B = torch.randn(*((batch, N, K) if transB else (batch, K, N)), device=dev, generator=g)
    Ad = (A.transpose(1, 2) if transA else A).double()
    Bd = (B.transpose(1, 2) if transB else B).double()
    ref = Ad @ Bd
    den = ref.abs().max().item()
    kw = dict(transA=transA, transB=transB, lda=A.shape[2], ldb=B.shape[2], ldc=N, batch=batch,
              strideA=A.shape[1] * A.shape[2], strideB=B.shape[1] * B.shape[2], strideC=M * N, splitk=splitk)
    for tile in tiles:
        out = []
        for prec in (ops.GEMM_F32, ops.GEMM_BF16X6):
            C = torch.full((batch, M, N), float("nan"), device=dev)
            for _ in range(3):
                ops.gemm(A, B, C, M, N, K, tile=tile, precision=prec, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.gemm(A, B, C, M, N, K, tile=tile, precision=prec, **kw)
            e1.record()
            e1.synchronize()
            t = e0.elapsed_time(e1) / reps * 1e-3
            err = (C.double() - ref).abs().max().item() / den
            out.append((t, err))
        fl = 2.0 * M * N * K * batch
        (t0, e0_), (t1, e1_) = out
        print(f"{name:26s} M={M:6d} N={N:5d} K={K:6d} b={batch} sk={splitk:3d} tile={tile:3d} | f32 {t0*1e6:7.1f} us "
              f"{fl/t0/1e12:6.1f} TF err {e0_:.2e} | bf16x6 {t1*1e6:7.1f} us {fl/t1/1e12:6.1f} TF err {e1_:.2e} | "
              f"x{t0/t1:.2f}", flush=True)


if __name__ == "__main__":
    run("NT odd (77)", 100, 35, 77, tiles=(64,))
    run("NN odd", 130, 260, 36, transB=False)
    run("TN odd", 130, 260, 36, transA=True, transB=False)
    run("TT odd", 131, 67, 45, transA=True, transB=True, tiles=(64,))
    run("NT MLP fwd", 8192, 512, 512)
    run("NN MLP dgrad", 8192, 512, 512, transB=False)
    run("TN MLP wgrad sk8", 512, 512, 8192, transA=True, transB=False, splitk=8)
    run("TN MLP wgrad sk16", 512, 512, 8192, transA=True, transB=False, splitk=16)
    run("NN mixer b4", 4608, 256, 256, transB=False, batch=4)
    run("NT mixer dgrad b4", 4608, 256, 256, transB=True, batch=4)
    run("TN mixer wgrad b4 sk9", 256, 256, 4608, transA=True, transB=False, batch=4, splitk=9)
    run("NN out-layer", 8192, 2048, 512, transB=False)
    run("NT out dgrad", 8192, 512, 2048)
    run("TN out wgrad sk4", 512, 2048, 8192, transA=True, transB=False, splitk=4)
    run("NN embed", 8192, 512, 360, transB=False)
    run("NT big", 81920, 512, 512)
    run("NT huge-K small vals", 1024, 1024, 8192, scale=1e-3)
    run("NT M-medium", 16384, 1024, 1024)
