"""Data-path transforms under the reference's PIPELINES names (mmdet/datasets/pipelines)."""
from .oa_mix import OAMix  # noqa: F401
from .device_pipeline import (Collect, Compose, DefaultFormatBundle, DevicePipeline, ImageToTensor,  # noqa: F401
                              MultiScaleFlipAug, Normalize, Pad, SyntheticCityscapes)
from .geometric import LoadAnnotations, LoadImageFromFile, RandomFlip, Resize  # noqa: F401
from .corrupt import Corrupt  # noqa: F401
