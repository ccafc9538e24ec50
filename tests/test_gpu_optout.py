"""The documented opt-out switches (DESIGN.md "Environment knobs") select the FALLBACK kernels of the product path.  They
are read once per process (static initialisers in csrc/, module-level reads in dpot_amd/), so each switch gets its own
child pytest process that re-runs a parity subset under it: the golden-vector model tests (test_gpu_model.py), the
DPOT-Tiny / -Small / -Medium gradient cases against the oracle in fp32 and bf16 channel-MLP mode, and - for the switches
that only act on DPOT-L's launch shapes - the DPOT-L batch-16 case against the reference's golden numbers; the `auto` GEMM
precision child also runs DPOT-S / -M at batch 32 against theirs (the bench lines of S / M / L run under it).  Keeps the
fallback kernels under the round-end GPU gate instead of a by-hand run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SMALL_SET = ("test_full_model_gradients_vs_oracle and (TINY-32 or SMALL-1) or "
             "test_bf16_channel_mlp_mode_vs_oracle and MEDIUM-1 or test_tiny_at_other_resolutions_vs_oracle and 64-3")
LARGE_SET = "test_vs_reference_golden and LARGE-16"

# switch group (set together in one child: they act on different kernels) -> (-k expression over test_gpu_sizes.py, also
# run the golden-vector model tests of test_gpu_model.py?).  Four children: the round-end GPU gate has to stay short.
SWITCHES = {
    # bf16 channel-MLP kernels.  Default since round 4: the B-direct kernels (W fragments straight into registers).
    # BD=0: the LDS-DMA kernels of rounds 2-3 in their own default selection (two-workgroup kernel, 128 x 192 tiles, pairs)
    # (DPOT_BF16P_ROWFORM=1, an opt-IN riding along: it only acts on the B-direct pair launch, which BD=0 switches off - so it
    # gets the DPOT-M case of its own below)
    # (DPOT_AFNO_WGRAD_GAUSS96=0 rides along: the 192 x 192 four-product AFNO weight-gradient kernel that gemm_tn96g_kernel
    # replaced at 96 channels per block - only DPOT-L has that shape)
    "DPOT_BF16P_BD=0 DPOT_AFNO_WGRAD_GAUSS96=0": (LARGE_SET, False),
    # round 5 defaults switched OFF: four-product AFNO weight gradients, the separate pack passes of the bf16 channel MLP (no packs
    # from the one-launch AFNO layer / the GroupNorm backward) - the forms these replaced stay under the gate
    "DPOT_AFNO_WGRAD_GAUSS=0 DPOT_GRAD_PACKS=0 DPOT_AFNO_LAYER_PACKS=0":
        ("test_vs_reference_golden and SMALL-32 or test_full_model_gradients_vs_oracle and TINY-32", False),
    # round 5 opt-ins that were built and rejected by measurement - kept under the gate: the one-launch AFNO layer BACKWARD and
    # the weight gradients on ROW-form operands through the transposing LDS read (DPOT-M at batch 32: reference golden, fp32
    # leg + bf16 leg; DPOT-S at batch 32 runs the one-launch layer forward + backward)
    "DPOT_AFNO_LAYER_BWD=1 DPOT_BF16P_ROWFORM=1": ("test_vs_reference_golden and (MEDIUM-32 or SMALL-32)", False),
    # not an opt-OUT but the mode `bench.py --config S|M|L|L20` runs: fp32 GEMMs >= 3 GFLOP on the fp32-accurate bf16x6 operand
    # split (`auto`).  The DPOT-L batch-16 reference golden (fp32 path at rtol 1e-4, then the bf16 channel-MLP mode) and the
    # Tiny / M gradient cases against the oracle must hold under it as they do with native fp32 MFMA
    "DPOT_GEMM_PRECISION=auto": (LARGE_SET + " or test_vs_reference_golden and (SMALL-32 or MEDIUM-32)"
                                 " or test_full_model_gradients_vs_oracle and TINY-32"
                                 " or test_bf16_channel_mlp_mode_vs_oracle and MEDIUM-1", False),
    # every round-3/4 fallback at once: four-product fused mixer, separate GroupNorm / DFT kernels, GroupNorm never applied
    # on load, generic GEMM instead of the panel / weight-gradient kernels, explicit patch matrix, three reduce launches per
    # block, eight layout launches, B-direct with eight waves everywhere / column-major order / 256-wide tiles / un-paired
    ("DPOT_AFNO_3MULT=0 DPOT_GN_DFT=0 DPOT_GN_ONLOAD=0 DPOT_PANEL_GEMM=0 DPOT_GEMM_TN=0 DPOT_EMBED_IMPLICIT=0 "
     "DPOT_BLOCK_FINALIZE=0 DPOT_LAYOUT_JOBS=0 DPOT_BF16P_BD_CPW=1 DPOT_BF16P_RASTER=0 DPOT_BF16P_ROWMAJOR=0 "
     "DPOT_BF16P_TILE192=0 DPOT_BF16P_PAIR=0"): (SMALL_SET, True),
    # the mixer as two generic GEMM launches, and the LDS-DMA bf16 kernels with the 12-wave kernel for every launch, 128 x 256
    # tiles only, un-paired weight gradients
    # (DPOT_AFNO_LAYER=0: the three launches per AFNO layer forward at the batches where `auto` picks the one-launch kernel)
    "DPOT_AFNO_FUSED=0 DPOT_AFNO_LAYER=0 DPOT_BF16P_BD=0 DPOT_BF16P_DUO=0 DPOT_BF16P_TILE192=0 DPOT_BF16P_PAIR=0":
        (SMALL_SET + " or test_vs_reference_golden and SMALL-32", True),
}


def _child_cmd(switch):
    expr, with_model = SWITCHES[switch]
    env = dict(os.environ)
    for kv in switch.split():
        k, v = kv.split("=")
        env[k] = v
    files = ["tests/test_gpu_sizes.py"] + (["tests/test_gpu_model.py"] if with_model else [])
    if with_model:
        expr = f"({expr}) or (test_gpu_model and not baseline_configs_forward)"   # (31 s of CPU oracle per child for DPOT-L)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + files + ["-k", expr]
    return cmd, env


@pytest.fixture(scope="module")
def children():
    """all child pytest processes are started TOGETHER (round 5: the five of them took 230 s of the GPU gate one after the other;
    most of a child's time is the CPU oracle and process start-up, the GPU work of all of them together is a few seconds) and
    each parametrised test below waits for its own"""
    procs = {}
    for switch in SWITCHES:
        cmd, env = _child_cmd(switch)
        env["OMP_NUM_THREADS"] = str(max(4, (os.cpu_count() or 8) // len(SWITCHES)))
        procs[switch] = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    yield procs
    for p in procs.values():
        if p.poll() is None:
            p.kill()


@pytest.mark.parametrize("switch", list(SWITCHES))
def test_parity_subset_under_opt_out_switch(switch, children):
    p = children[switch]
    try:
        out, _ = p.communicate(timeout=1500)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    tail = out[-3000:]
    assert p.returncode == 0, f"{switch}: parity subset failed\n{tail}"
    last = [ln for ln in out.splitlines() if " passed" in ln]
    assert last and " failed" not in last[-1], tail
    print(f"[{switch}] {last[-1].strip()}")
