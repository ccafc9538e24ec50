"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the drop-in model keeps the
reference's state_dict layout, host logic (LR schedule, sharding) is right, and the product path refuses CPU
tensors instead of silently falling back."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import dpot_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from dpot_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built_lib):
    from dpot_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dpot_hip.h")).read()
    declared = set(re.findall(r"\b(dpot_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dpot_stream_t"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES.keys()), declared ^ set(_lib.SIGNATURES.keys())
    lib = _lib.load()                       # binds every symbol or raises AttributeError
    assert lib.dpot_version() >= 100
    assert lib.dpot_colsum_parts(8192) == 32
    assert lib.dpot_gemm_auto_splitk(81920, 512, 512, 1) == 1
    assert lib.dpot_gemm_auto_splitk(512, 512, 8192, 1) > 1


def test_gemm_desc_layout_matches_c_struct(built_lib):
    """sizeof / field offsets of the ctypes mirror vs the C struct (compiled with the host compiler)"""
    import ctypes
    import subprocess
    import tempfile
    from dpot_amd._lib import GemmDesc
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "dpot_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",' \
          'sizeof(dpot_gemm_desc),offsetof(dpot_gemm_desc,strideA),offsetof(dpot_gemm_desc,aux),' \
          'offsetof(dpot_gemm_desc,res_mod),offsetof(dpot_gemm_desc,tile));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    want = [ctypes.sizeof(GemmDesc), GemmDesc.strideA.offset, GemmDesc.aux.offset, GemmDesc.res_mod.offset,
            GemmDesc.tile.offset]
    assert got == want


def test_state_dict_layout_matches_reference():
    from dpot_amd import DPOTNet
    for kw in (R.MINI, dict(R.MINI, normalize=True), dict(R.MINI, time_agg="mlp"), R.TINY):
        cfg = R.DPOTConfig(**kw)
        m = DPOTNet(**kw)
        sd = m.state_dict()
        shapes = R.param_shapes(cfg)
        assert list(sd.keys()) == list(shapes.keys())
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), k
        m.load_state_dict(R.recipe_state_dict(cfg))           # a reference-shaped checkpoint loads unchanged
        # per-component loading as in utils/utilities.py:112-166
        m.patch_embed.load_state_dict({k[len("patch_embed."):]: v for k, v in sd.items()
                                       if k.startswith("patch_embed.")})
        m.blocks[0].load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items() if k.startswith("blocks.0.")})
    assert sum(p.numel() for p in DPOTNet(**R.TINY).parameters()) == 7534643


def test_constructor_defaults_and_asserts():
    import inspect
    from dpot_amd import DPOTNet
    sig = inspect.signature(DPOTNet.__init__)
    want = dict(img_size=224, patch_size=16, mixing_type='afno', in_channels=1, out_channels=4, in_timesteps=1,
                out_timesteps=1, n_blocks=4, embed_dim=768, out_layer_dim=32, depth=12, modes=32, mlp_ratio=1.,
                n_cls=12, normalize=False, act='gelu', time_agg='exp_mlp')
    got = {k: v.default for k, v in sig.parameters.items() if k != "self"}
    assert got == want and list(got) == list(want)
    with pytest.raises(AssertionError):
        DPOTNet(img_size=32, patch_size=8, embed_dim=40, n_blocks=3, depth=1)      # width % num_blocks
    with pytest.raises(KeyError):
        DPOTNet(img_size=32, patch_size=8, embed_dim=32, depth=1, act="nope")


def test_no_cpu_fallback():
    from dpot_amd import DPOTNet, _lib
    m = DPOTNet(**R.MINI)
    with pytest.raises(_lib.DpotHipError):
        m(torch.zeros(1, 32, 32, 4, 3))
    from dpot_amd.functional import rel_l2_loss
    with pytest.raises(_lib.DpotHipError):
        rel_l2_loss(torch.zeros(1, 4, 4, 1, 2), torch.zeros(1, 4, 4, 1, 2))


def test_product_code_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dpot_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"


def test_one_cycle_lr_matches_torch():
    from dpot_amd.train import one_cycle_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    total = 200
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, div_factor=1e4, pct_start=0.2,
                                              final_div_factor=1e4, total_steps=total)
    for s in range(total):
        assert abs(opt.param_groups[0]["lr"] - one_cycle_lr(s, total, 1e-3, 0.2)) <= 1e-12 + 1e-9 * 1e-3, s
        opt.step()
        if s + 1 < total:
            sch.step()


def test_shard_indices_partition():
    from dpot_amd.dp import shard_indices
    W, n, bs = 4, 103, 5
    per_rank = [shard_indices(n, bs, r, W, epoch=3, seed=7) for r in range(W)]
    assert len({len(p) for p in per_rank}) == 1
    flat = [i for p in per_rank for b in p for i in b]
    assert len(flat) == len(set(flat))
    assert all(len(b) == bs for p in per_rank for b in p)
    assert per_rank[0] == shard_indices(n, bs, 0, W, epoch=3, seed=7)
    assert per_rank[0] != shard_indices(n, bs, 0, W, epoch=4, seed=7)


def test_flat_params_views():
    from dpot_amd import DPOTNet
    from dpot_amd.train import FlatParams
    m = DPOTNet(**R.MINI)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m)
    assert fp.names[-6:] == [f"cls_head.{i}.{k}" for i in (0, 2, 4) for k in ("weight", "bias")]
    assert fp.n_head < fp.total and fp.n_head % 4 == 0
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    for p, o in zip(fp.params, fp.offsets):
        assert p.data_ptr() == fp.flat.data_ptr() + 4 * o
        assert p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o
    m.load_state_dict(R.recipe_state_dict(R.DPOTConfig(**R.MINI)))       # in-place: views stay attached
    assert fp.params[0].data_ptr() == fp.flat.data_ptr() + 4 * fp.offsets[0]
