"""MI355X-native block-matching motion estimation (hot path of ashish-nr/BlockBasedMotionEstimation).

The compute path is hand-written HIP for gfx950 behind the C-ABI of include/bbme.h
(libbbme.so, built in-tree by `python -m blockbasedmotionestimation_amd.build`).
MF and Flow mirror the reference's classes; there is no CPU fallback.
"""
from ._capi import BbmeError, LIB_PATH  # noqa: F401
from .motion_framework import MF, MFBatch, plan_padding, pad_zero, pyr_down, resize_x4  # noqa: F401
from .rw_flow import Flow, FlowWriter, subsample_div4  # noqa: F401
from .synth import synth_pair, warp_pair_from_flow  # noqa: F401

__all__ = ["MF", "MFBatch", "Flow", "FlowWriter", "BbmeError", "plan_padding", "pad_zero", "pyr_down", "resize_x4",
           "subsample_div4", "synth_pair", "warp_pair_from_flow"]
