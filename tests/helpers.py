"""shared helpers for the parity tests"""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: outputs match the reference PyTorch-CPU path within rtol=1e-4 (fp32 / complex64).
# Element-wise rtol is meaningless for values that cancel to ~0, so - like torch.testing - we pair it with an
# absolute term scaled to the tensor's magnitude:  |a-b| <= RTOL*|b| + RTOL*max|b|
RTOL = 1e-4


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def assert_close(a, b, what="", rtol=RTOL, atol_scale=RTOL):
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a.detach().cpu()).double()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b.detach().cpu()).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    scale = b.abs().max().item()
    tol = rtol * b.abs() + atol_scale * scale + 1e-30
    err = (a - b).abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; "
                           f"max|d|={err.max().item():.3e} scale={scale:.3e}")
    return err.max().item() / (scale + 1e-30)


def assert_sub(t, fx, key, what="", rtol=RTOL):
    """compare a big tensor against a stored subsample + checksums"""
    stride = int(fx[key + ".stride"])
    f = t.detach().cpu().reshape(-1)
    assert_close(f[::stride], fx[key + ".sub"], what + ".sub", rtol=rtol)
    s = f.double().sum().item()
    a = f.double().abs().sum().item()
    assert abs(a - float(fx[key + ".abssum"])) <= 10 * rtol * float(fx[key + ".abssum"]), what + ".abssum"
    assert abs(s - float(fx[key + ".sum"])) <= 10 * rtol * float(fx[key + ".abssum"]), what + ".sum"


def set_tune(monkeypatch, **kv):
    """DPOT_TUNE with the given keys set (merged over whatever the process - e.g. an opt-out child - already has); only the
    PYTHON-side readers see a change made inside a running process (the C library reads DPOT_TUNE once): use it for the keys
    dpot_amd/ops.py consults per call (mixer, afno_layer, packs, fused_small, embed_implicit)"""
    import os
    cur = dict(x.split("=") for x in os.environ.get("DPOT_TUNE", "").split(",") if x)
    cur.update({k: str(v) for k, v in kv.items()})
    monkeypatch.setenv("DPOT_TUNE", ",".join(f"{k}={v}" for k, v in cur.items()))


def tune_value(key, default):
    import os
    cur = dict(x.split("=") for x in os.environ.get("DPOT_TUNE", "").split(",") if x)
    return int(cur.get(key, default))

