"""MI355X-native per-frame recurrent inference path of the embodied object detector.

Public surface (mirrors what `Detic/train_mp3d.py --eval-only` touches):
    setup_cfg / get_cfg            config (yacs-compatible CfgNode, `_BASE_`, KEY VALUE overrides)
    build_model(cfg[, state_dict]) registries + CustomRCNNRecurrent on the HIP kernels
    model([[frame, ...]])          -> [{"instances": Instances}, ...]
The arithmetic lives in libeod_hip.so (hand-written gfx950 kernels behind the C ABI of include/eod_hip.h).
"""
import os as _os

# The schedule uses up to ~8 streams of one priority (4 per model + one per scene of a BatchedSequences).  The HIP runtime maps the
# streams of a priority onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin, and two streams that share a hardware queue
# run strictly one after the other (tools/experiments/chain_contention.hip; two scenes in lock-step: 182 frames/s with 4 queues,
# 204 with 8).  Read by the runtime when it initialises, i.e. at the first GPU call: importing this package first is enough.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import CfgNode, get_cfg, setup_cfg  # noqa: E402,F401
from .registry import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY,  # noqa: F401
                       build_model)
from .structures import Boxes, Instances  # noqa: F401

__version__ = "0.1.0"
