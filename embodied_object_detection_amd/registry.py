"""The four registries `build_model(cfg)` resolves, with the reference's names
(`Detic/configs/Base-C2_L_R5021k_640b64_4x_recurrent.yaml:2,5,8,16`; registration sites
`custom_rcnn.py:333`, `timm.py:507`, `centernet.py:30`, `detic_roi_heads.py:29`)."""
from __future__ import annotations

from typing import Callable, Dict


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._map: Dict[str, Callable] = {}

    def register(self, obj=None, name: str = None):
        def deco(o):
            key = name or o.__name__
            assert key not in self._map, f"{key} already registered in {self._name}"
            self._map[key] = o
            return o
        return deco if obj is None else deco(obj)

    def get(self, name: str):
        if name not in self._map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry! Registered: {list(self._map)}")
        return self._map[name]

    def __contains__(self, name):
        return name in self._map


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")


def build_model(cfg, state_dict=None):
    """`detectron2.modeling.build_model(cfg)` for this path: resolves MODEL.META_ARCHITECTURE and builds it on
    MODEL.DEVICE.  `state_dict` (reference-keyed) supplies the weights; None -> MODEL.WEIGHTS is loaded, or the
    deterministic synthetic weights when that is empty."""
    from . import modeling  # noqa: F401  (registers the classes)
    meta = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    return meta(cfg, state_dict)
