"""MI355X-native (gfx950) CNN backbone + temporal Transformer encoder hot path."""
from .modules import (ConvBNReLUPool, HybridCrossEntropyLoss, MultiheadAttention, TransformerCNNHybrid,  # noqa: F401
                      TransformerEncoder)
from .optim import HybridAdamW  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401
from .fct import FCT, DiceLoss  # noqa: F401
from .encoder32k import Bottleneck, Encoder_32K  # noqa: F401
from .clips import ClipCSVDataset, ClipPipeline, SyntheticClipSource, collate_clips, t_major  # noqa: F401

__all__ = ["TransformerCNNHybrid", "TransformerEncoder", "MultiheadAttention", "ConvBNReLUPool", "HybridCrossEntropyLoss", "HybridAdamW", "GraphedTrainStep", "FCT", "DiceLoss", "Bottleneck", "Encoder_32K", "ClipCSVDataset", "ClipPipeline", "SyntheticClipSource", "collate_clips", "t_major"]
