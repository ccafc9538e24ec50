#!/usr/bin/env python
"""does a forked branch of tiny kernels hide behind big kernels inside a hipGraph?  (cost of fork/join edges)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dpot_amd import ops

M, N, K = 8192, 512, 512
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
Cs = [torch.empty(M, N, device="cuda") for _ in range(2)]
small = [torch.randn(4096, device="cuda") for _ in range(3)]
side = torch.cuda.Stream()
NIT = 20


def body(mode):
    cur = torch.cuda.current_stream()
    for i in range(NIT):
        ops.gemm(A, W, Cs[i & 1], M, N, K, transB=True, lda=K, ldb=K, ldc=N, splitk=1)
        if mode == "serial":
            small[2].copy_(small[0])
        elif mode == "fork":
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                small[2].copy_(small[0])
    if mode == "fork":
        cur.wait_stream(side)


def run(mode):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body(mode); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body(mode)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e6


for mode in ("none", "serial", "fork"):
    if mode == "serial":
        def add_(a, b, out): out.copy_(a)
    print(mode, f"{run(mode):.1f} us per graph of {NIT} GEMMs", flush=True)
