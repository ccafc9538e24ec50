#!/usr/bin/env python
"""micro-benchmark of dpot_gemm_f32 on the GPU box: time per launch / TFLOP/s for the DPOT GEMM shapes"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import ops  # noqa: E402


def bench(name, M, N, K, transA=False, transB=True, tile=0, batch=1, epi=None, splitk=1, reps=30):
    dev = "cuda"
    A = torch.randn(batch, K, M, device=dev) if transA else torch.randn(batch, M, K, device=dev)
    B = torch.randn(batch, N, K, device=dev) if transB else torch.randn(batch, K, N, device=dev)
    C = torch.empty(batch, M, N, device=dev)
    kw = dict(transA=transA, transB=transB, lda=A.shape[2], ldb=B.shape[2], ldc=N, batch=batch,
              strideA=A.shape[1] * A.shape[2], strideB=B.shape[1] * B.shape[2], strideC=M * N, tile=tile,
              splitk=splitk)
    if epi == "gelu":
        bias = torch.randn(batch, N, device=dev)
        pre = torch.empty_like(C)
        kw.update(bias=bias, strideBias=N, act=1, mode=ops.EPI_ACT, preact=pre, ldpre=N, stridePre=M * N)
    elif epi == "dgelu":
        aux = torch.randn(batch, M, N, device=dev)
        kw.update(act=1, mode=ops.EPI_DACT, aux=aux, ldaux=N, strideAux=M * N)
    elif epi == "res":
        res = torch.randn(batch, M, N, device=dev)
        bias = torch.randn(batch, N, device=dev)
        kw.update(bias=bias, strideBias=N, res=res, ldres=N, strideRes=M * N)
    for _ in range(3):
        ops.gemm(A, B, C, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(A, B, C, M, N, K, **kw)
    e1.record()
    e1.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    fl = 2.0 * M * N * K * batch
    print(f"{name:44s} M={M:6d} N={N:5d} K={K:5d} b={batch} tile={tile:3d} sk={splitk:3d} epi={str(epi):6s} "
          f"{t*1e6:8.1f} us  {fl/t/1e12:6.1f} TF", flush=True)


if __name__ == "__main__":
    for tile in (128, 64):
        for K in (128, 256, 512, 1024, 2048, 5120):
            bench("NT MLP-like", 8192, 512, K, tile=tile)
    for tile in (128, 64):
        bench("NT big (patch2-like M)", 81920, 512, 512, tile=tile)
        bench("NT gelu epi", 8192, 512, 512, tile=tile, epi="gelu")
        bench("NT res epi", 8192, 512, 512, tile=tile, epi="res")
        bench("NN dgrad dgelu", 8192, 512, 512, transB=False, tile=tile, epi="dgelu")
        bench("NN timeagg fwd", 8192, 512, 5120, transB=False, tile=tile)
        bench("NN out-layer", 8192, 2048, 512, transB=False, tile=tile, epi="gelu")
        bench("mixer NN batch4", 4608, 256, 256, transB=False, tile=tile, batch=4, epi="gelu")
        bench("mixer NN batch4 noepi", 4608, 256, 256, transB=False, tile=tile, batch=4)
    for tile, sk in ((64, 8), (128, 16), (128, 8), (64, 4), (64, 16)):
        bench("TN wgrad 512x512", 512, 512, 8192, transA=True, transB=False, tile=tile, splitk=sk)
    for tile, sk in ((64, 1), (128, 1), (128, 2), (64, 2)):
        bench("TN wgrad timeagg", 5120, 512, 8192, transA=True, transB=False, tile=tile, splitk=sk)
    for tile, sk in ((64, 2), (128, 1), (128, 4), (64, 4)):
        bench("TN wgrad out0", 512, 2048, 8192, transA=True, transB=False, tile=tile, splitk=sk)
    bench("skinny tail fwd", 524288, 32, 32, tile=64, epi="gelu")
    bench("skinny tail dgrad", 524288, 32, 32, transB=False, tile=64, epi="dgelu")
    bench("skinny wgrad", 32, 32, 524288, transA=True, transB=False, tile=64, splitk=512)
    bench("patch gemm1", 81920, 36, 448, tile=64, epi="gelu")
    bench("patch gemm2", 81920, 512, 36, tile=128, epi="res")
