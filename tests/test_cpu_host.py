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
    from dpot_amd.dp import dp_steps_per_epoch, shard_indices
    W, n, bs = 4, 103, 5
    per_rank = [shard_indices(n, bs, r, W, epoch=3, seed=7) for r in range(W)]
    assert len({len(p) for p in per_rank}) == 1
    flat = [i for p in per_rank for b in p for i in b]
    assert set(flat) == set(range(n))                         # nothing is dropped ...
    assert len(flat) == 6 * W * bs                            # ... the last round wraps around (21 batches -> 24)
    assert all(len(b) == bs for p in per_rank for b in p)
    assert dp_steps_per_epoch(n, bs, W) == (6, 21) and len(per_rank[0]) == 6
    assert per_rank[0] == shard_indices(n, bs, 0, W, epoch=3, seed=7)
    assert per_rank[0] != shard_indices(n, bs, 0, W, epoch=4, seed=7)


def test_shard_batches_equal_accelerate_batch_sampler_shard():
    """the reference's sharding IS accelerate's `BatchSamplerShard(split_batches=False, even_batches=True)` over a
    `BatchSampler(drop_last=False)` (train_temporal_parallel.py:102,120,185); accelerate is installed in this image, so
    compare with the class itself on ragged, exact and degenerate sizes"""
    accelerate = pytest.importorskip("accelerate")
    from accelerate.data_loader import BatchSamplerShard
    from torch.utils.data import BatchSampler
    from dpot_amd.dp import dp_steps_per_epoch, shard_batches
    rng = np.random.default_rng(5)
    for n, bs, W in [(103, 5, 4), (100, 5, 4), (101, 5, 4), (97, 8, 8), (64, 8, 8), (65, 8, 8), (7, 4, 4), (3, 2, 8),
                     (1, 4, 2), (40, 4, 1), (41, 4, 1), (1000, 32, 8), (1023, 32, 8), (29, 3, 2)]:
        order = rng.permutation(n).tolist()
        for r in range(W):
            want = list(BatchSamplerShard(BatchSampler(order, bs, drop_last=False), num_processes=W, process_index=r,
                                          split_batches=False, even_batches=True))
            got = shard_batches(order, bs, r, W)
            assert got == want, (n, bs, W, r)
            assert len(got) == dp_steps_per_epoch(n, bs, W)[0]


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


def test_dp_lr_rule_matches_accelerate_stepping():
    """train_temporal_parallel.py:150,185: OneCycleLR sized by the UNSHARDED loader, stepped `world` times per optimiser
    step by accelerate's AcceleratedScheduler (split_batches=False) - dp.dp_one_cycle_lr gives the lr each optimiser
    step actually uses"""
    from dpot_amd.dp import dp_one_cycle_lr
    world, total, max_lr, pct = 4, 240, 1e-3, 0.25
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=max_lr)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, div_factor=1e4, pct_start=pct,
                                                final_div_factor=1e4, total_steps=total)
    for step in range(total // world):
        used = opt.param_groups[0]["lr"]                     # the lr optimizer.step() sees at this update
        assert abs(dp_one_cycle_lr(step, world, total, max_lr, pct_start=pct) - used) <= 1e-9 * max_lr + 1e-12, step
        opt.step()
        for _ in range(world):                               # AcceleratedScheduler.step: num_processes steps
            if sched.last_epoch < total - 1:
                sched.step()


def test_flat_params_execution_order_and_stage_buckets():
    """FlatParams lays parameters out by forward execution stage (time_agg before the blocks, out_layer last before the
    cls_head tail); BucketedGradReducer cuts buckets only at stage boundaries"""
    from dpot_amd import DPOTNet
    from dpot_amd.dp import BucketedGradReducer
    from dpot_amd.train import FlatParams
    m = DPOTNet(**dict(R.MINI, depth=4))
    fp = FlatParams(m)
    stages = [BucketedGradReducer._stage(n) for n in fp.names]
    head = [s for s in stages if s >= 0]
    assert head == sorted(head) and stages[-1] == -1 and fp.names[0].startswith("patch_embed.")
    assert fp.names.index("time_agg_layer.w") < fp.names.index("blocks.0.norm1.weight")
    red = BucketedGradReducer(fp, n_buckets=8)
    for k in range(red.n_buckets):
        st = {red.stage_of[i] for i in red.members[k]}
        nxt = {red.stage_of[i] for i in red.members[k + 1]} if k + 1 < red.n_buckets else set()
        assert not (st & nxt), "a stage must not straddle two buckets"
    assert red.tail_bucket == red.n_buckets - 1


def test_load_components_warns_for_absent_known_component():
    import warnings
    from dpot_amd import DPOTNet
    from dpot_amd.infer import load_components_from_pretrained
    cfg = R.DPOTConfig(**R.MINI)
    m = DPOTNet(**R.MINI)                                    # normalize=False: no scale_feats
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        load_components_from_pretrained(m, R.recipe_state_dict(cfg, salt=1), ["blocks", "scale_feats"])
    assert any("scale_feats" in str(x.message) for x in w)


def test_dpot_tune_switchboard(built_lib, monkeypatch):
    """round 6: ONE environment variable (DPOT_TUNE="key=val,...") behind every fallback selector.  The Python side parses it per
    call and rejects unknown keys; the C side (dpot_tune) parses the same string once per process - checked in a child process
    because this one may already have cached it"""
    import subprocess
    import sys
    from dpot_amd import ops
    monkeypatch.delenv("DPOT_TUNE", raising=False)
    assert all(ops.tune(k) == d for k, (d, _) in ops.TUNE_KEYS.items())
    assert len(ops.TUNE_KEYS) <= 12
    monkeypatch.setenv("DPOT_TUNE", "mixer=4, afno_layer=1,packs=0")
    assert (ops.tune("mixer"), ops.tune("afno_layer"), ops.tune("packs"), ops.tune("panel")) == (4, 1, 0, 1)
    monkeypatch.setenv("DPOT_TUNE", "no_such_key=1")
    with pytest.raises(ValueError):
        ops.tune("mixer")
    code = ("from dpot_amd import _lib; l = _lib.load(); "
            "print(l.dpot_tune(b'bf16p_bd', 1), l.dpot_tune(b'panel', 1), l.dpot_tune(b'wgrad_gauss', 1), l.dpot_tune(b'pan', 7))")
    env = dict(os.environ, DPOT_TUNE="bf16p_bd=0,wgrad_gauss=0")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-400:]
    assert out.stdout.split() == ["0", "1", "0", "7"], out.stdout
    # no other DPOT_* variable is read by the package or the library any more (bench.py has its own DPOT_BENCH_* / DPOT_DP_*)
    allowed = {"DPOT_TUNE", "DPOT_HIP_LIB", "DPOT_GEMM_PRECISION", "DPOT_MLP_PRECISION"}
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dpot_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for m in re.finditer(r"(?:getenv\(|environ(?:\.get)?[\[(])\s*\"(DPOT_[A-Z0-9_]+)\"", src):
                    assert m.group(1) in allowed, (f, m.group(1))


def test_adam_pack_plan_tables(built_lib):
    """host side of dpot_adam_step_packs (ops.AdamPackPlan): the channel-MLP weights of a flat parameter buffer become 64 x 256
    tile jobs, everything else plain ranges of <= 2^20 elements; jobs + ranges cover [0, n_active) exactly once"""
    import ctypes
    from dpot_amd import _lib, ops

    class FakePacks:
        bf16, planes = True, 1

    E, mh = 256, 512
    sizes = [1000, mh * E, 40, E * mh, (1 << 20) + 5000, mh * E, 12]       # small | W1 | small | W2 | long gap | W1' | tail
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += (n + 3) // 4 * 4
    flat = torch.zeros(off)
    view = lambda i, shape: flat[offs[i]:offs[i] + sizes[i]].view(shape)
    w1, w2, w1b = view(1, (mh, E)), view(3, (E, mh)), view(5, (mh, E))
    pp = FakePacks()
    pp.jobs = [(w1, mh, E, E, False), (w1, E, mh, E, True), (w2, E, mh, mh, False), (w2, mh, E, mh, True),
               (w1b, mh, E, E, False), (w1b, E, mh, E, True)]
    pp.n = len(pp.jobs)
    pp.bufs = [torch.zeros(r * k, dtype=torch.bfloat16) for _, r, k, _, _ in pp.jobs]
    n_active = offs[6]                                       # the tail tensor is not updated
    plan = ops.AdamPackPlan.build(flat, n_active, pp)
    assert plan is not None and plan.ntiles == 2 * (mh // 64) * (E // 256) + (E // 64) * (mh // 256)
    tab = (_lib.AdamPackJob * 3).from_buffer_copy(plan.jobs_dev.numpy().tobytes())
    covered = torch.zeros(n_active, dtype=torch.int32)
    tiles = 0
    for j in tab:
        assert j.tile0 == tiles and j.R % 64 == 0 and j.K % 256 == 0
        tiles += (j.R // 64) * (j.K // 256)
        covered[j.off:j.off + j.R * j.K] += 1
    assert tiles == plan.ntiles == plan.tile_job_dev.numel()
    assert plan.tile_job_dev.tolist() == sorted(plan.tile_job_dev.tolist())
    rg = plan.ranges_dev.view(-1, 2).tolist()
    assert len(rg) == plan.nranges and max(l for _, l in rg) == plan.max_range <= ops.AdamPackPlan.MAX_RANGE
    for st, ln in rg:
        covered[st:st + ln] += 1
    assert int(covered.min()) == 1 and int(covered.max()) == 1           # every active element exactly once
    # a weight that does not tile (K % 256 != 0) disables the plan
    bad = torch.zeros(64 * 200).view(64, 200)
    pp2 = FakePacks()
    pp2.jobs = [(bad, 64, 200, 200, False), (bad, 200, 64, 200, True)]
    pp2.n, pp2.bufs = 2, [torch.zeros(64 * 200, dtype=torch.bfloat16)] * 2
    assert ops.AdamPackPlan.build(bad.view(-1), bad.numel(), pp2) is None
