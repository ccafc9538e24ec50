"""Host logic of dpot_amd.infer: checkpoint / component loading contracts of utils/utilities.py:99-166 (CPU only:
parameters can be constructed and loaded without a GPU; forward on CPU raises by design)."""
from collections import OrderedDict

import pytest
import torch

from oracle import dpot_ref as R


def _model(salt=None):
    from dpot_amd import DPOTNet
    m = DPOTNet(**R.MINI)
    if salt is not None:
        m.load_state_dict(R.recipe_state_dict(R.DPOTConfig(**R.MINI), salt=salt))
    return m


def test_load_model_from_checkpoint_plain_ddp_prefix_and_file(tmp_path):
    from dpot_amd.infer import load_model_from_checkpoint
    cfg = R.DPOTConfig(**R.MINI)
    sd = R.recipe_state_dict(cfg, salt=3)
    for variant in (sd, OrderedDict(("module." + k, v) for k, v in sd.items()), {"model": sd, "args": None}):
        m = _model()
        load_model_from_checkpoint(m, variant)
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd[k]), k
    path = tmp_path / "model_mini.pth"
    torch.save({"model": OrderedDict(("module." + k, v) for k, v in sd.items()), "optimizer": {}}, path)
    m = _model()
    load_model_from_checkpoint(m, str(path))
    assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    bad = OrderedDict(sd)
    bad.pop("pos_embed")
    with pytest.raises(RuntimeError):
        load_model_from_checkpoint(_model(), bad)


def test_load_components_from_pretrained_subsets():
    from dpot_amd.infer import COMPONENTS, load_components_from_pretrained
    cfg = R.DPOTConfig(**R.MINI)
    src = R.recipe_state_dict(cfg, salt=7)
    prefixes = {"patch_embed": "patch_embed.", "pos": "pos_embed", "blocks": "blocks.", "time_agg": "time_agg_layer.",
                "cls_head": "cls_head.", "out": "out_layer."}
    for comp, prefix in prefixes.items():
        m = _model(salt=1)
        before = {k: v.clone() for k, v in m.state_dict().items()}
        pos_obj = m.pos_embed
        load_components_from_pretrained(m, OrderedDict(("module." + k, v) for k, v in src.items()), [comp])
        assert m.pos_embed is pos_obj                                  # copied in place: flat bindings stay valid
        for k, v in m.state_dict().items():
            want = src[k] if k.startswith(prefix) else before[k]
            assert torch.equal(v, want), (comp, k)
    m = _model(salt=1)
    load_components_from_pretrained(m, src, "all")
    assert all(torch.equal(v, src[k]) for k, v in m.state_dict().items())
    with pytest.raises(KeyError):
        load_components_from_pretrained(_model(), src, ["decoder"])
    assert set(prefixes) | {"scale_feats"} == set(COMPONENTS)
