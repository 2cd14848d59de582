"""Registers the hot-path classes under the reference's registry names."""
from . import backbone, centernet, roi_heads, meta_arch  # noqa: F401
from .meta_arch import CustomRCNNRecurrent  # noqa: F401
