"""Config surface of the hot path: a yacs/detectron2-compatible `CfgNode` without detectron2.

Mirrors what `setup()` does at `Detic/train_mp3d.py:661-683`: `get_cfg()` + `add_centernet_config`
(`Detic/third_party/CenterNet2/centernet/config.py:3-88`) + `add_detic_config` (`Detic/detic/config.py:4-200`),
`merge_from_file` with the `_BASE_` chain, then trailing `KEY VALUE` overrides.  Only the keys the inference
path reads are given defaults; unknown keys found in a YAML are accepted (the reference YAMLs carry training
keys this path never looks at).
"""
from __future__ import annotations

import ast
import copy
import os
from typing import Any, List

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self) -> "CfgNode":
        return copy.deepcopy(self)

    def merge_from_other(self, other: dict):
        for k, v in other.items():
            if k == "_BASE_":
                continue
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k].merge_from_other(v)
            else:
                self[k] = _coerce(self.get(k), v)

    def merge_from_file(self, path: str):
        with open(path, "r") as f:
            data = yaml.safe_load(f) or {}
        base = data.get("_BASE_")
        if base:
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(path), base)
            self.merge_from_file(base)
        self.merge_from_other(data)

    def merge_from_list(self, opts: List[Any]):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = _coerce(node.get(parts[-1]), val)

    def freeze(self):
        return self


def _coerce(old, new):
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    return new


def get_cfg() -> CfgNode:
    """Defaults of every key the per-frame inference path reads (values of d2 / centernet / detic configs)."""
    C = CfgNode
    cfg = C({
        "VERSION": 2,
        "OUTPUT_DIR": "./output",
        "FP16": False,
        "WITH_IMAGE_LABELS": False,
        "MODEL": {
            "META_ARCHITECTURE": "GeneralizedRCNN",
            "DEVICE": "cuda",
            "WEIGHTS": "",
            "MASK_ON": False,
            "PIXEL_MEAN": [103.530, 116.280, 123.675],
            "PIXEL_STD": [1.0, 1.0, 1.0],
            "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 2},
            "FPN": {"IN_FEATURES": [], "OUT_CHANNELS": 256, "NORM": "", "FUSE_TYPE": "sum"},
            "TIMM": {"BASE_NAME": "resnet50", "OUT_LEVELS": (3, 4, 5), "NORM": "FrozenBN", "FREEZE_AT": 0, "PRETRAINED": False},
            "PROPOSAL_GENERATOR": {"NAME": "RPN", "MIN_SIZE": 0},
            "CENTERNET": {
                "NUM_CLASSES": 80, "IN_FEATURES": ["p3", "p4", "p5", "p6", "p7"], "FPN_STRIDES": [8, 16, 32, 64, 128],
                "PRIOR_PROB": 0.01, "INFERENCE_TH": 0.05, "CENTER_NMS": False, "NMS_TH_TRAIN": 0.6, "NMS_TH_TEST": 0.6,
                "PRE_NMS_TOPK_TRAIN": 1000, "POST_NMS_TOPK_TRAIN": 100, "PRE_NMS_TOPK_TEST": 1000, "POST_NMS_TOPK_TEST": 100,
                "NORM": "GN", "USE_DEFORMABLE": False, "NUM_CLS_CONVS": 4, "NUM_BOX_CONVS": 4, "NUM_SHARE_CONVS": 0,
                "WITH_AGN_HM": False, "ONLY_PROPOSAL": False, "AS_PROPOSAL": False, "NOT_NMS": False, "NOT_NORM_REG": True,
                # training keys (centernet/config.py:24-44): target assignment and losses of `modeling/training.py`
                "LOC_LOSS_TYPE": "giou", "SIGMOID_CLAMP": 1e-4, "HM_MIN_OVERLAP": 0.8, "MIN_RADIUS": 4,
                "SOI": [[0, 80], [64, 160], [128, 320], [256, 640], [512, 10000000]], "POS_WEIGHT": 1.0, "NEG_WEIGHT": 1.0,
                "REG_WEIGHT": 2.0, "HM_FOCAL_BETA": 4, "HM_FOCAL_ALPHA": 0.25, "LOSS_GAMMA": 2.0, "IGNORE_HIGH_FP": -1.0,
                "MORE_POS": False, "NO_REDUCE": False,
            },
            "ROI_HEADS": {
                "NAME": "Res5ROIHeads", "NUM_CLASSES": 80, "IN_FEATURES": ["res4"], "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5,
                "IOU_THRESHOLDS": [0.5], "MASK_WEIGHT": 1.0, "ONE_CLASS_PER_PROPOSAL": False,
                # detectron2 defaults read by the training forward (label_and_sample_proposals)
                "BATCH_SIZE_PER_IMAGE": 512, "POSITIVE_FRACTION": 0.25, "PROPOSAL_APPEND_GT": True,
            },
            "ROI_BOX_HEAD": {
                "NAME": "", "NUM_FC": 0, "FC_DIM": 1024, "NUM_CONV": 0, "CONV_DIM": 256, "POOLER_RESOLUTION": 14,
                "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2", "CLS_AGNOSTIC_BBOX_REG": False,
                "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0),
                "USE_ZEROSHOT_CLS": False, "ZEROSHOT_WEIGHT_PATH": "datasets/metadata/lvis_v1_clip_a+cname.npy",
                "ZEROSHOT_WEIGHT_DIM": 512, "NORM_WEIGHT": True, "NORM_TEMP": 50.0, "IGNORE_ZERO_CATS": False, "USE_BIAS": 0.0,
                "MULT_PROPOSAL_SCORE": False, "USE_SIGMOID_CE": False, "PRIOR_PROB": 0.01, "ADD_FEATURE_TO_PROP": False,
                "ADD_IMAGE_BOX": False, "IMAGE_BOX_SIZE": 1.0, "WS_NUM_PROPS": 128,
                "SMOOTH_L1_BETA": 0.0, "BBOX_REG_LOSS_TYPE": "smooth_l1", "BBOX_REG_LOSS_WEIGHT": 1.0, "USE_FED_LOSS": False,
            },
            "ROI_BOX_CASCADE_HEAD": {
                "BBOX_REG_WEIGHTS": ((10.0, 10.0, 5.0, 5.0), (20.0, 20.0, 10.0, 10.0), (30.0, 30.0, 15.0, 15.0)),
                "IOUS": (0.5, 0.6, 0.7),
            },
            "ROI_MASK_HEAD": {
                "NAME": "MaskRCNNConvUpsampleHead", "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 0, "NUM_CONV": 0,
                "CONV_DIM": 256, "NORM": "", "CLS_AGNOSTIC_MASK": False, "POOLER_TYPE": "ROIAlignV2",
            },
            # detic config.py:56-74 (memory keys)
            "MAP_MERGE_TYPE": "", "MAP_FEAT_FUSION": "", "FREEZE_BACKBONE": False, "UNFROZEN_LAYERS": [],
            "MEMORY_FEATURE_WEIGHT": 100, "TEST_SAVE_SEMMAP": False, "SEMMAP_PATH": "", "MEMORY_TYPE": "",
            "MEMORY_CLS_SCORE_THRESH": 0.3, "MEMORY_OBS_SCORE_THRESH": 0.4, "MAP_FEATURE_WEIGHT": 500,
            "TEST_DATA_PATH": "embodied_data/mp3d_example/", "TRAIN_DATA_PATH": "embodied_data/mp3d_example/",
            "MEMORY_PATH": "embodied_data/mp3d_example/memory_data", "TEST_TYPE": "default",
            "RESET_CLS_TESTS": False, "TEST_CLASSIFIERS": [], "TEST_NUM_CLASSES": [],
            "DYNAMIC_CLASSIFIER": False, "WITH_CAPTION": False, "SYNC_CAPTION_BATCH": False, "CAP_BATCH_RATIO": 4,
            "DATASET_LOSS_WEIGHT": [],
        },
        "INPUT": {"FORMAT": "BGR", "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "CUSTOM_AUG": "", "TEST_SIZE": 640,
                  "NOT_CLAMP_BOX": False},
        "DATASETS": {"TRAIN": (), "TEST": ()},
        # NUM_WORKERS_TRAIN_MP3D: the MP3D training DataLoader's worker processes (hard-coded 2 in train_mp3d.py:566; 0 = in-process)
        "DATALOADER": {"NUM_WORKERS": 4, "SAMPLER_TRAIN": "TrainingSampler", "NUM_WORKERS_TRAIN_MP3D": 2},
        "TEST": {"DETECTIONS_PER_IMAGE": 100, "EVAL_PERIOD": 0},
        # detectron2's SOLVER defaults + Detic's additions (detic/config.py:153-157): read by solver.py / modeling/training.py
        "SOLVER": {
            "LR_SCHEDULER_NAME": "WarmupMultiStepLR", "MAX_ITER": 40000, "BASE_LR": 0.001, "MOMENTUM": 0.9, "NESTEROV": False,
            "WEIGHT_DECAY": 0.0001, "WARMUP_FACTOR": 0.001, "WARMUP_ITERS": 1000, "WARMUP_METHOD": "linear", "IMS_PER_BATCH": 16,
            "CHECKPOINT_PERIOD": 5000, "TRAIN_ITER": -1,
            "CLIP_GRADIENTS": {"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0},
            "USE_CUSTOM_SOLVER": False, "OPTIMIZER": "SGD", "BACKBONE_MULTIPLIER": 1.0, "CUSTOM_MULTIPLIER": 1.0,
            "CUSTOM_MULTIPLIER_NAME": [],
        },
        "DEBUG": False,
        "VIS_THRESH": 0.3,
        "EVAL_PROPOSAL_AR": False,
    })
    return cfg


def config_path(name: str) -> str:
    """Path of a YAML shipped with this package (same file names as `Detic/configs/`)."""
    p = os.path.join(CONFIG_DIR, name)
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p


DEFAULT_CONFIG = "Detic_LCOCOI21k_CLIP_R5021k_640b32_4x_ft4x_max-size_mp3d_recurrent.yaml"


def setup_cfg(config_file: str = None, opts: List[Any] = None) -> CfgNode:
    """`setup(args)` of `Detic/train_mp3d.py:661-683` minus logging."""
    cfg = get_cfg()
    cfg.merge_from_file(config_file or config_path(DEFAULT_CONFIG))
    if opts:
        cfg.merge_from_list(list(opts))
    zs = cfg.MODEL.ROI_BOX_HEAD.ZEROSHOT_WEIGHT_PATH
    if not os.path.isabs(zs) and not os.path.exists(zs):
        # 'datasets/metadata/mp3d_clip.npy' is resolved relative to the Detic checkout in the reference; fall back to
        # the copy of the 20 KB fixture shipped with the package
        cand = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metadata", os.path.basename(zs))
        if os.path.exists(cand):
            cfg.MODEL.ROI_BOX_HEAD.ZEROSHOT_WEIGHT_PATH = cand
    return cfg
