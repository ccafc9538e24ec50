"""world_size-2 `gloo` tests (CPU) of the data-parallel path: bucketed all-reduce overlapped via autograd hooks,
parameter broadcast, and the DDP averaging convention (golden g8 numbers generated from the reference)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from dpot_amd import DPOTNet
    from dpot_amd.dp import BucketedGradReducer, all_reduce_scalar
    from dpot_amd.train import FlatParams
    from oracle import dpot_ref as R

    cfg = R.DPOTConfig(**R.MINI)
    sd = R.recipe_state_dict(cfg, salt=17)
    # the parameter container of the product model (its HIP forward cannot run on CPU; the oracle computes the
    # per-rank gradients here, exactly as the reference model would under DDP)
    model = DPOTNet(**R.MINI)
    if rank == 0:
        model.load_state_dict(sd)                       # rank 1 starts from its own random init
    fp = FlatParams(model)
    red = BucketedGradReducer(fp, n_buckets=3, overlap=True)
    red.broadcast_parameters(0)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), f"broadcast mismatch on {k}"
    assert red.n_buckets >= 3 and red.ranges[-1][0] == fp.n_head     # cls_head tail has its own bucket

    B = 4
    xx = R.recipe_input((B, cfg.img_size, cfg.img_size, cfg.in_timesteps, cfg.in_channels), salt=81)
    yy = R.recipe_input((B, cfg.img_size, cfg.img_size, 1, cfg.out_channels), salt=82)
    msk = torch.ones(B, cfg.img_size, cfg.img_size, 1, cfg.out_channels)
    sl = slice(2 * rank, 2 * rank + 2)

    fp.zero_grad()
    red.begin_step()
    params = {k: p for k, p in model.named_parameters()}
    pred, _ = R.dpot_forward(params, xx[sl], cfg)
    loss = R.rel_l2_loss(pred, yy[sl], msk[sl])
    loss.backward()                                     # hooks fire per parameter -> buckets launch as they complete
    launched_in_backward = sum(red._launched)
    red.finish()
    assert launched_in_backward >= red.n_buckets - 1    # everything but the grad-less cls_head tail overlapped
    g = fp.grad * red.grad_scale                        # DDP average
    total = all_reduce_scalar(loss.detach().clone())
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), gnorm=np.float64(torch.sqrt((g.double() ** 2).sum()).item()),
                 names=np.array(fp.names),
                 norms=np.array([g[o:o + p.numel()].norm().item() for p, o in zip(fp.params, fp.offsets)]),
                 loss_sum=np.float64(total.item()))
    # all ranks must hold identical reduced gradients
    ref = g.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(ref, g)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_bucketed_allreduce(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "dp.npz"))
    fx = load("g8_dp")
    assert abs(float(got["gnorm"]) - float(fx["grad_norm"])) <= 1e-4 * float(fx["grad_norm"])
    want = dict(zip([str(n) for n in fx["names"]], fx["grad_norms"]))
    for n, v in zip(got["names"], got["norms"]):
        assert abs(v - want[str(n)]) <= 1e-4 * want[str(n)] + 1e-7, n


def test_bucket_count_follows_the_gradient_bytes():
    """round 5: the default bucket count comes from the gradient BYTES (~32 MB per bucket, 2..8; a fixed 4 before) -
    DPOT-Tiny 30 MB -> 2 (+ the grad-less cls_head tail bucket), DPOT-M 489 MB / DPOT-L 2 GB -> 8"""
    from dpot_amd.dp import auto_n_buckets
    assert auto_n_buckets(30 << 20) == 2 and auto_n_buckets(1 << 20) == 2
    assert auto_n_buckets(123 << 20) == 4
    assert auto_n_buckets(489 << 20) == 8 and auto_n_buckets(2 << 30) == 8
