"""dpot_amd - MI355X-native (gfx950 / CDNA4) implementation of the DPOT auto-regressive forward/backward step.

    from dpot_amd import DPOTNet            # drop-in for models/dpot.py::DPOTNet of HaoZhongkai/DPOT
    from dpot_amd.train import FlatParams, FusedAdam, train_step, GraphedTrainStep
    from dpot_amd.dp import BucketedGradReducer

The compute path is libdpot_hip.so (hand-written HIP kernels behind the C ABI in include/dpot_hip.h).
"""
from .model import DPOTNet  # noqa: F401
from . import _lib  # noqa: F401

__version__ = "0.2.6"
__all__ = ["DPOTNet"]
