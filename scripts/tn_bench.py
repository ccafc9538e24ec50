"""micro-benchmark: the fused weight-gradient launches of csrc/gemm_tn.hip at the DPOT-Tiny B=32 shapes, by split factor;
'warm' = the same operands every launch (they stay in the 256 MB Infinity Cache), 'cold' = rotating through operand sets
that total more than it (what the train step sees)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops


def timeit(fns, reps=24):
    for f in fns: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fns[i % len(fns)]()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def afno(Mm, nb, bs, E, nsets):
    sets = []
    for _ in range(nsets):
        sets.append([torch.randn(Mm, 2 * E, device="cuda") for _ in range(4)])
    N = 2 * bs
    dw1, dw2 = torch.empty(2, nb, bs, bs, device="cuda"), torch.empty(2, nb, bs, bs, device="cuda")
    db1, db2 = torch.empty(2, nb, bs, device="cuda"), torch.empty(2, nb, bs, device="cuda")
    fl = 2.0 * 2 * nb * N * N * Mm
    auto = ops.afno_wgrad2_splitk(Mm, nb, bs)
    for sk in (2, 4, 5, 6, 8, 10, 12, 16, 18, 24, 36):
        nslab = Mm // 32
        sps = (nslab + sk - 1) // sk
        if sps * (sk - 1) >= nslab: continue
        fns = [(lambda s=s, sk=sk: ops.afno_wgrad2(s[0], s[1], s[2], s[3], nb, bs, dw1, db1, dw2, db2, sk)) for s in sets]
        t = timeit(fns)
        print(f"  afno_wgrad2 Mm={Mm} nb={nb} bs={bs} sets={nsets} splitk={sk:2d}{'*' if sk == auto else ' '} {t*1e6:7.1f} us  {fl/t/1e12:6.1f} TF", flush=True)


def mlp(T, E, mh, nsets):
    sets = [[torch.randn(T, E, device="cuda"), torch.randn(T, mh, device="cuda"), torch.randn(T, E, device="cuda"),
             torch.randn(T, mh, device="cuda")] for _ in range(nsets)]
    dW2, dW1 = torch.empty(E, mh, device="cuda"), torch.empty(mh, E, device="cuda")
    db2, db1 = torch.empty(E, device="cuda"), torch.empty(mh, device="cuda")
    fl = 2.0 * 2 * T * E * mh
    auto = ops.mlp_wgrad2_splitk(T, E, mh)
    for sk in (2, 4, 8, 16):
        fns = [(lambda s=s, sk=sk: ops.mlp_wgrad2(s[0], s[1], s[2], s[3], dW2, db2, dW1, db1, sk)) for s in sets]
        t = timeit(fns)
        print(f"  mlp_wgrad2 T={T} E={E} mh={mh} sets={nsets} splitk={sk:2d}{'*' if sk == auto else ' '} {t*1e6:7.1f} us  {fl/t/1e12:6.1f} TF", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "L":
    # DPOT-Large at 256 x 256, batch 16: 32 x 17 kept modes per sample, 16 blocks of 96 channels (DPOT_AFNO_WGRAD_GAUSS96=0/1)
    print(f"DPOT-L B=16 AFNO weight gradients, DPOT_AFNO_WGRAD_GAUSS96={os.environ.get('DPOT_AFNO_WGRAD_GAUSS96', '1')}")
    afno(16 * 32 * 17, 16, 96, 1536, 3)
    sys.exit(0)
print("DPOT-Tiny B=32 (kernel + reduce launch per call)")
for nsets in (1, 5):
    afno(4608, 4, 128, 512, nsets)
print("DPOT-S / -M B=32 AFNO weight gradients")
afno(4608, 8, 128, 1024, 3)
print("DPOT-Tiny channel MLP")
for nsets in (1, 5):
    mlp(8192, 512, 512, nsets)
