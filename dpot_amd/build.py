"""Build libdpot_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the library is a plain
C-ABI shared object (include/dpot_hip.h) loaded through ctypes."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdpot_hip.so")
SOURCES = ["core.hip", "gemm.hip", "gemm_panel.hip", "gemm_tn.hip", "gemm_bf16p.hip", "afno_mlp.hip", "afno_mlp6.hip", "afno_fused.hip", "dft.hip", "norm.hip", "gn_dft.hip", "misc.hip", "loss_opt.hip", "tail.hip", "data.hip", "embed.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]
# per-source extras.  afno_mlp6: the operand split runs beside MFMAs, where the v_pk_add_f32 the SLP vectoriser forms out of
# neighbouring subtractions are slower than the scalar instructions (MI355X_MICROARCH.md; measured -3 % on the launch)
EXTRA_FLAGS = {"afno_mlp6.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 kernels)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "dft_fast.h"), os.path.join(CSRC, "gemm_split.h"), os.path.join(CSRC, "gemm_epi.h"), os.path.join(CSRC, "gemm_bf16p_common.h"), os.path.join(os.path.dirname(HERE), "include", "dpot_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[dpot_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
