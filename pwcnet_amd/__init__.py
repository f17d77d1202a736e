"""pwcnet_amd -- MI355X-native (gfx950) PWC-Net inference path.

Drop-in for the forward pass of daigo0927/pwcnet (model.py / modules.py op-level API):
Python host classes over hand-written HIP kernels in csrc/ (C ABI: include/pwc_hip.h).
"""
from .model import PWCDCNet  # noqa: F401
from .pipeline import ForwardPipeline  # noqa: F401
from .modules import (ContextNetwork, CostVolumeLayer, FeaturePyramidExtractor_custom,  # noqa: F401
                      OpticalFlowEstimator_custom, WarpingLayer, resize_bilinear)

__all__ = ["PWCDCNet", "ForwardPipeline", "FeaturePyramidExtractor_custom", "WarpingLayer", "CostVolumeLayer",
           "OpticalFlowEstimator_custom", "ContextNetwork", "resize_bilinear"]
