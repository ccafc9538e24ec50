"""same-box yardstick for the bf16 channel-MLP GEMMs: this build's launches beside torch.mm (hipBLASLt) on bf16 tensors at
the DPOT-S / -M / -L shapes (bench.gemm_yardstick; measurement only - nothing here is on the product path).
usage: python scripts/gemm_yardstick.py [S] [M] [L16] [L8]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

SHAPES = {"S": (8192, 1024, 1024), "M": (8192, 1024, 4096), "L16": (16384, 1536, 6144), "L8": (8192, 1536, 6144)}
which = [a for a in sys.argv[1:] if a in SHAPES] or ["M", "L16"]
print(f"torch {torch.__version__}, preferred BLAS backend: {torch.backends.cuda.preferred_blas_library()}")
for name in which:
    M, E, mh = SHAPES[name]
    rows = bench.gemm_yardstick(M, E, mh)
    print(f"== {name}: tokens {M}, E {E}, hidden {mh} ({2.0 * M * E * mh / 1e9:.1f} GFLOP per product)")
    print(f"  {'form':<42s} {'ours us':>8s} {'TF':>7s} | {'hipBLASLt bf16-out':>18s} {'f32-out':>8s} {'best TF':>8s} | ours/blaslt (bf16-out, f32-out)")
    for r in rows:
        print(f"  {r['form']:<42s} {r['ours_us']:8.1f} {r['ours_TF']:7.1f} | {r['blaslt_bf16out_us']:18.1f} "
              f"{(r['blaslt_f32out_us'] or float('nan')):8.1f} {r['blaslt_best_TF']:8.1f} | {r['ours_over_blaslt_bf16out']:.3f}  {r['ours_over_blaslt_f32out']}")
    print(json.dumps({"shape": name, "rows": rows}))
    torch.cuda.empty_cache()
