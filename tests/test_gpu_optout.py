"""The documented fallback selectors (DPOT_TUNE keys: dpot_amd/ops.py TUNE_KEYS) select the FALLBACK kernels
of the product path.  The C library reads DPOT_TUNE once per process, so each group of keys gets its own
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
    # bf16p_bd=0: the LDS-DMA kernels of rounds 2-3 for every launch, in their own default selection (128 x 192 tiles, pairs)
    # (wgrad_gauss=0 rides along: four-product AFNO weight gradients - at DPOT-L's 96 channels per block that is the 192 x 192
    # kernel that gemm_tn96g_kernel replaced)
    "DPOT_TUNE=bf16p_bd=0,wgrad_gauss=0": (LARGE_SET, False),
    # round 5 / 6 defaults switched OFF: four-product AFNO weight gradients, no bf16 packs from the one-launch AFNO layer / the
    # GroupNorm backward / Adam (the separate pack passes these replaced stay under the gate)
    "DPOT_TUNE=wgrad_gauss=0,packs=0":
        ("test_vs_reference_golden and SMALL-32 or test_full_model_gradients_vs_oracle and TINY-32", False),
    # not an opt-OUT but the mode `bench.py --config S|M|L|L20` runs: fp32 GEMMs >= 3 GFLOP on the fp32-accurate bf16x6 operand
    # split (`auto`).  The DPOT-L batch-16 reference golden (fp32 path at rtol 1e-4, then the bf16 channel-MLP mode) and the
    # Tiny / M gradient cases against the oracle must hold under it as they do with native fp32 MFMA
    "DPOT_GEMM_PRECISION=auto": (LARGE_SET + " or test_vs_reference_golden and (SMALL-32 or MEDIUM-32)"
                                 " or test_full_model_gradients_vs_oracle and TINY-32"
                                 " or test_bf16_channel_mlp_mode_vs_oracle and MEDIUM-1", False),
    # every round-3/4 fallback at once: four-product fused mixer, separate GroupNorm / DFT kernels, GroupNorm never applied
    # on load, generic GEMM instead of the panel / weight-gradient kernels, explicit patch matrix, three reduce launches per
    # block, eight layout launches, B-direct with eight waves everywhere / column-major order / 256-wide tiles / un-paired,
    # no pack-both path
    "DPOT_TUNE=mixer=4,gn_fuse=0,panel=0,embed_implicit=0,fused_small=0,bf16p_shape=0,pack_both=0": (SMALL_SET, True),
    # the mixer as two generic GEMM launches, the three launches per AFNO layer forward at the batches where `auto` picks the
    # one-launch kernel, and the LDS-DMA bf16 kernels for every launch, 128 x 256 tiles only, un-paired weight gradients
    "DPOT_TUNE=mixer=0,afno_layer=0,bf16p_bd=0,bf16p_shape=0":
        (SMALL_SET + " or test_vs_reference_golden and SMALL-32", True),
}


def _child_cmd(switch):
    expr, with_model = SWITCHES[switch]
    env = dict(os.environ)
    for kv in switch.split():
        k, _, v = kv.partition("=")
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
