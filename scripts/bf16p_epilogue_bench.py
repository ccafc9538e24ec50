import os, sys
sys.path.insert(0, "/root/repo")
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
for M, N, K in ((4096, 6144, 1536), (4096, 1536, 6144), (8192, 4096, 1024), (8192, 1024, 4096)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    pb = ops.PanelPacks([(W, N, K, K, False)], bf16=True); pb.refresh()
    Ap = ops.bf16_pack_rows(A)
    X = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * N * K
    out = torch.empty(M, N, device="cuda")
    r = {}
    r["linear"] = timeit(lambda: ops.gemm_bf16p(Ap, pb.bufs[0], M, N, K, bias=b, out=out))
    r["act+pre"] = timeit(lambda: ops.gemm_bf16p(Ap, pb.bufs[0], M, N, K, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True, out=out))
    r["act+pre+packs(no C)"] = timeit(lambda: ops.gemm_bf16p_packed(Ap, pb.bufs[0], M, N, K, bias=b, act=1, mode=ops.EPI_ACT, save_pre=True, pack_rows=True, pack_trans=True, store=False))
    r["dact packs+colsum(no C)"] = timeit(lambda: ops.gemm_bf16p_packed(Ap, pb.bufs[0], M, N, K, act=1, mode=ops.EPI_DACT, aux=X, pack_rows=True, pack_trans=True, colsum=True, store=False))
    print(M, N, K, " | ".join(f"{k}: {v:.1f} us {fl/v/1e6:.0f} TF" for k, v in r.items()), flush=True)
