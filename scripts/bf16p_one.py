"""ONE form of the bf16 channel-MLP GEMMs at a DPOT shape, launched 12 times (for rocprofv3 --pmc passes):
python scripts/bf16p_one.py M fc1_fwd|fc2_fwd|fc2_dgrad|fc1_dgrad|pair"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpot_amd import ops

shapes = {"S": (8192, 1024, 1024), "M": (8192, 1024, 4096), "L16": (16384, 1536, 6144), "L4": (4096, 1536, 6144)}
M, E, mh = shapes[sys.argv[1]]
form = sys.argv[2]
x = torch.randn(M, E, device="cuda"); do = torch.randn(M, E, device="cuda")
W1 = torch.randn(mh, E, device="cuda") * 0.03; W2 = torch.randn(E, mh, device="cuda") * 0.03
b1 = torch.randn(mh, device="cuda") * 0.1; b2 = torch.randn(E, device="cuda") * 0.1
pk = ops.PanelPacks([(W1, mh, E, E, False), (W1, E, mh, E, True), (W2, E, mh, mh, False), (W2, mh, E, mh, True)], bf16=True)
pk.refresh()
xp, xpT, _ = ops.bf16_pack_both(x)
dop, dopT, _ = ops.bf16_pack_both(do)
_, D, hp, hpT, _ = ops.gemm_bf16p_packed(xp, pk.bufs[0], M, mh, E, bias=b1, act=1, mode=ops.EPI_ACT, save_dact=True,
                                         pack_rows=True, pack_trans=True, store=False)
_, _, dhp, dhpT, _ = ops.gemm_bf16p_packed(dop, pk.bufs[3], M, mh, E, act=1, mode=ops.EPI_DACT, dact=D, pack_rows=True,
                                           pack_trans=True, colsum=True, store=False)
o0, o1 = torch.empty(E, mh, device="cuda"), torch.empty(mh, E, device="cuda")
fns = {
    "fc1_fwd": lambda: ops.gemm_bf16p_packed(xp, pk.bufs[0], M, mh, E, bias=b1, act=1, mode=ops.EPI_ACT, save_dact=True,
                                             pack_rows=True, pack_trans=True, store=False),
    "fc2_fwd": lambda: ops.gemm_bf16p(hp, pk.bufs[2], M, E, mh, bias=b2, res=x),
    "fc2_dgrad": lambda: ops.gemm_bf16p_packed(dop, pk.bufs[3], M, mh, E, act=1, mode=ops.EPI_DACT, dact=D, pack_rows=True,
                                               pack_trans=True, colsum=True, store=False),
    "fc1_dgrad": lambda: ops.gemm_bf16p(dhp, pk.bufs[1], M, E, mh),
    "pair": lambda: ops.gemm_bf16p_pair(dopT, hpT, E, mh, dhpT, xpT, mh, E, M, out0=o0, out1=o1),
}
torch.cuda.synchronize()
print("MARK begin", flush=True)
for _ in range(12):
    fns[form]()
torch.cuda.synchronize()
