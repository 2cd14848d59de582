"""MI355X-native per-frame recurrent inference path of the embodied object detector.

Public surface (mirrors what `Detic/train_mp3d.py --eval-only` touches):
    setup_cfg / get_cfg            config (yacs-compatible CfgNode, `_BASE_`, KEY VALUE overrides)
    build_model(cfg[, state_dict]) registries + CustomRCNNRecurrent on the HIP kernels
    model([[frame, ...]])          -> [{"instances": Instances}, ...]
The arithmetic lives in libeod_hip.so (hand-written gfx950 kernels behind the C ABI of include/eod_hip.h).
"""
from .config import CfgNode, get_cfg, setup_cfg  # noqa: F401
from .registry import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY,  # noqa: F401
                       build_model)
from .structures import Boxes, Instances  # noqa: F401

__version__ = "0.1.0"
