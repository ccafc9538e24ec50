"""Where does the in-step / back-to-back gap of the AFNO mixer launch come from (DPOT-Tiny 50 vs 44 us, DPOT-M 95 vs 84 us)?
The same launch in a hipGraph of 48 launches, (a) same weights + same spectrum every launch (bench.py's roofline figure),
(b) weights cycling over 12 layers (cold in L2), same spectrum, (c) same weights, spectrum cycling over 12 buffers (each larger
than... the L2s), (d) both cycling, (e) as (d) with a producer-like kernel (a copy of the spectrum) in front of every launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dpot_amd import ops

def run(name, E, nb, B):
    bs = E // nb
    Mm = B * 16 * 9
    NL = 12
    Ss = [torch.randn(Mm, 2 * E, device="cuda") for _ in range(NL)]
    layers = []
    for _ in range(NL):
        wc1 = torch.randn(2, nb, bs, bs, device="cuda") * 0.05; wc2 = torch.randn(2, nb, bs, bs, device="cuda") * 0.05
        bc1 = torch.randn(2, nb, bs, device="cuda") * 0.1; bc2 = torch.randn(2, nb, bs, device="cuda") * 0.1
        layers.append((wc1, bc1)); layers.append((wc2, bc2))
    pk = ops.AfnoPacks(layers)
    items = pk.refresh()
    tmp = torch.empty_like(Ss[0])

    def launch(li, si, producer=False):
        (_, b1, W1f, _), (_, b2, W2f, _) = items[2 * li], items[2 * li + 1]
        if producer:
            tmp.copy_(Ss[si]); src = tmp
        else:
            src = Ss[si]
        return ops.afno_mlp2(src, W1f, b1, W2f, b2, nb, bs, 1, mode=0, want_pre=True, want_mid=False, layout=1)

    def timeit(fn, reps=48):
        for i in range(3): fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps): fn(i)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    t_copy = timeit(lambda i: tmp.copy_(Ss[i % NL]))
    print(f"{name}: spectrum {Mm} x {2 * E} ({Mm * 2 * E * 4 / 1e6:.1f} MB), weights per layer pair {2 * 2 * nb * bs * bs * 4 * 0.75 / 1e6:.2f} MB")
    print(f"  (a) same weights, same spectrum        {timeit(lambda i: launch(0, 0)):7.2f} us")
    print(f"  (b) weights cycle, same spectrum       {timeit(lambda i: launch(i % NL, 0)):7.2f} us")
    print(f"  (c) same weights, spectrum cycles      {timeit(lambda i: launch(0, i % NL)):7.2f} us")
    print(f"  (d) both cycle                         {timeit(lambda i: launch(i % NL, i % NL)):7.2f} us")
    print(f"  (e) both cycle + producer copy in front {timeit(lambda i: launch(i % NL, i % NL, True)) - t_copy:7.2f} us (copy {t_copy:.2f} us subtracted)", flush=True)

run("DPOT-Tiny B=32", 512, 4, 32)
run("DPOT-S/M B=32", 1024, 8, 32)
