"""Checkpoint surface of the hot path.

Two producers of a *reference-keyed* state dict (the names of SURVEY.md §8a, i.e. the
module tree behind `Detic/detic/modeling/meta_arch/custom_rcnn.py:334` built by
`build_p67_timm_fpn_backbone_recurrent` `Detic/detic/modeling/backbone/timm.py:507-531`):

* `load_checkpoint(path)`  - a detectron2 style `.pth` (`torch.load` -> `{'model': {...}}` or a bare
  state dict), what `DetectionCheckpointer.resume_or_load` reads at
  `Detic/train_mp3d.py:717-719`.  Tensors whose shape differs from the expected one are skipped
  and reported, missing keys are reported (d2 semantics).
* `synthetic_state_dict(seed)` - deterministic random weights of the same architecture; there
  is no network for the real `.pth`, so bench/tests use this and say so ("data": "synthetic").

All tensors are torch-layout fp32 CPU tensors (conv OIHW, linear [out,in], ConvTranspose
[in,out,kh,kw]); the device-side re-layout (BN folding, NHWC, K-major GEMM panels) happens in
`modeling/*` when a model is built.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

_METADATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metadata")
DEFAULT_ZS_WEIGHT = os.path.join(_METADATA_DIR, "mp3d_clip.npy")

RESNET50_LAYERS = (3, 4, 6, 3)
RESNET50_PLANES = (64, 128, 256, 512)
NUM_CASCADE = 3


def expected_shapes(num_classes: int = 20) -> "OrderedDict[str, Tuple[int, ...]]":
    """Name -> shape of every tensor the hot path consumes, in module order."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def bn(prefix, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}.{k}"] = (c,)

    base = "backbone.bottom_up.base"
    s[f"{base}.conv1.weight"] = (64, 3, 7, 7)
    bn(f"{base}.bn1", 64)
    inplanes = 64
    for li, (nblk, planes) in enumerate(zip(RESNET50_LAYERS, RESNET50_PLANES), start=1):
        for b in range(nblk):
            p = f"{base}.layer{li}.{b}"
            s[f"{p}.conv1.weight"] = (planes, inplanes, 1, 1)
            bn(f"{p}.bn1", planes)
            s[f"{p}.conv2.weight"] = (planes, planes, 3, 3)
            bn(f"{p}.bn2", planes)
            s[f"{p}.conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(f"{p}.bn3", planes * 4)
            if b == 0:
                s[f"{p}.downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                bn(f"{p}.downsample.1", planes * 4)
            inplanes = planes * 4
    for lvl, cin in ((3, 512), (4, 1024), (5, 2048)):
        s[f"backbone.fpn_lateral{lvl}.weight"] = (256, cin, 1, 1)
        s[f"backbone.fpn_lateral{lvl}.bias"] = (256,)
        s[f"backbone.fpn_output{lvl}.weight"] = (256, 256, 3, 3)
        s[f"backbone.fpn_output{lvl}.bias"] = (256,)
    for n in ("p6", "p7"):
        s[f"backbone.top_block.{n}.weight"] = (256, 256, 3, 3)
        s[f"backbone.top_block.{n}.bias"] = (256,)
    for i in (1, 2, 3):
        s[f"backbone.map_merge_projection{i}.weight"] = (256, 512, 1, 1)
        s[f"backbone.map_merge_projection{i}.bias"] = (256,)

    h = "proposal_generator.centernet_head"
    for i in range(4):
        s[f"{h}.bbox_tower.{3 * i}.weight"] = (256, 256, 3, 3)
        s[f"{h}.bbox_tower.{3 * i}.bias"] = (256,)
        s[f"{h}.bbox_tower.{3 * i + 1}.weight"] = (256,)
        s[f"{h}.bbox_tower.{3 * i + 1}.bias"] = (256,)
    s[f"{h}.bbox_pred.weight"] = (4, 256, 3, 3)
    s[f"{h}.bbox_pred.bias"] = (4,)
    s[f"{h}.agn_hm.weight"] = (1, 256, 3, 3)
    s[f"{h}.agn_hm.bias"] = (1,)
    for l in range(5):
        s[f"{h}.scales.{l}.scale"] = (1,)

    for k in range(NUM_CASCADE):
        s[f"roi_heads.box_head.{k}.fc1.weight"] = (1024, 256 * 7 * 7)
        s[f"roi_heads.box_head.{k}.fc1.bias"] = (1024,)
        s[f"roi_heads.box_head.{k}.fc2.weight"] = (1024, 1024)
        s[f"roi_heads.box_head.{k}.fc2.bias"] = (1024,)
        p = f"roi_heads.box_predictor.{k}"
        s[f"{p}.cls_score.linear.weight"] = (512, 1024)
        s[f"{p}.cls_score.linear.bias"] = (512,)
        s[f"{p}.cls_score.zs_weight"] = (512, num_classes + 1)
        s[f"{p}.bbox_pred.0.weight"] = (1024, 1024)
        s[f"{p}.bbox_pred.0.bias"] = (1024,)
        s[f"{p}.bbox_pred.2.weight"] = (4, 1024)
        s[f"{p}.bbox_pred.2.bias"] = (4,)
    m = "roi_heads.mask_head"
    for i in range(1, 5):
        s[f"{m}.mask_fcn{i}.weight"] = (256, 256, 3, 3)
        s[f"{m}.mask_fcn{i}.bias"] = (256,)
    s[f"{m}.deconv.weight"] = (256, 256, 2, 2)
    s[f"{m}.deconv.bias"] = (256,)
    s[f"{m}.predictor.weight"] = (1, 256, 1, 1)
    s[f"{m}.predictor.bias"] = (1,)
    return s


def load_zs_weight(path: str = DEFAULT_ZS_WEIGHT) -> torch.Tensor:
    """CLIP text matrix as the classifier holds it: [512, C+1], unit columns, zero bg column.

    Follows `Detic/detic/modeling/roi_heads/zero_shot_classifier.py:41-49` (load [C,512], transpose,
    append a zero column, L2-normalise columns).  Pure host arithmetic on a 20 KB constant.
    """
    w = torch.tensor(np.load(path), dtype=torch.float32).permute(1, 0).contiguous()
    w = torch.cat([w, w.new_zeros((w.shape[0], 1))], dim=1)
    return torch.nn.functional.normalize(w, p=2, dim=0)


def synthetic_state_dict(seed: int = 0, num_classes: int = 20,
                         zs_weight_path: str = DEFAULT_ZS_WEIGHT) -> Dict[str, torch.Tensor]:
    """Deterministic random weights for every tensor of `expected_shapes`.

    He-normal convs/linears; FrozenBN statistics are random (so BN folding is exercised) with the
    last BN of every bottleneck damped to keep the 16-block residual trunk at O(1) activations;
    head output layers get a logit spread of O(1) so the data-dependent selections
    (top-k / NMS / thresholds) are not decided by near-ties.
    """
    g = torch.Generator().manual_seed(seed)
    shapes = expected_shapes(num_classes)
    sd: Dict[str, torch.Tensor] = OrderedDict()

    def randn(shape, std):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std

    def rand(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    for name, shape in shapes.items():
        leaf = name.rsplit(".", 1)[1]
        if ".bn" in name or ".downsample.1." in name:
            damp = 0.25 if (".bn3." in name or ".downsample.1." in name) else 1.0
            if leaf == "weight":
                t = rand(shape, 0.8, 1.2) * damp
            elif leaf == "bias":
                t = randn(shape, 0.05)
            elif leaf == "running_mean":
                t = randn(shape, 0.05)
            else:
                t = rand(shape, 0.8, 1.2)
        elif name.endswith("zs_weight"):
            t = load_zs_weight(zs_weight_path)
            assert tuple(t.shape) == shape, (t.shape, shape)
        elif ".scales." in name:
            t = rand(shape, 0.9, 1.1)
        elif "bbox_tower" in name and len(shape) == 1:
            # GroupNorm affine
            t = rand(shape, 0.8, 1.2) if leaf == "weight" else randn(shape, 0.05)
        elif leaf == "bias":
            if name.endswith("agn_hm.bias"):
                t = torch.full(shape, -2.0)
            elif name.endswith("centernet_head.bbox_pred.bias"):
                t = torch.full(shape, 4.0)
            else:
                t = randn(shape, 0.02)
        else:
            fan_in = int(np.prod(shape[1:]))
            if name.endswith("deconv.weight"):
                fan_in = shape[0]
            std = math.sqrt(2.0 / fan_in)
            if name.endswith("agn_hm.weight"):
                std = 1.5 / math.sqrt(fan_in)
            elif name.endswith("centernet_head.bbox_pred.weight"):
                std = 2.0 / math.sqrt(fan_in)
            elif ".bbox_pred.2.weight" in name:
                std = 0.5 / math.sqrt(fan_in)
            elif "map_merge_projection" in name:
                std = 0.02 / math.sqrt(fan_in)
            elif name.endswith("mask_head.predictor.weight"):
                std = 2.0 / math.sqrt(fan_in)
            t = randn(shape, std)
        sd[name] = t.contiguous()
    return sd


def load_checkpoint(path: str, num_classes: int = 20, verbose: bool = True):
    """Read a d2 `.pth` and return (state_dict, report).

    report = {'missing': [...], 'shape_mismatch': [...], 'unexpected': [...]}.  Mirrors the
    non-strict load of `DetectionCheckpointer` used at `Detic/train_mp3d.py:717-719`.
    """
    obj = torch.load(path, map_location="cpu", weights_only=False)
    model = obj["model"] if isinstance(obj, dict) and "model" in obj else obj
    shapes = expected_shapes(num_classes)
    sd: Dict[str, torch.Tensor] = OrderedDict()
    report = {"missing": [], "shape_mismatch": [], "unexpected": []}
    for name, shape in shapes.items():
        if name not in model:
            report["missing"].append(name)
            continue
        t = model[name]
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        t = t.detach().to(torch.float32).cpu().contiguous()
        if tuple(t.shape) != tuple(shape):
            report["shape_mismatch"].append((name, tuple(t.shape), tuple(shape)))
            continue
        sd[name] = t
    for name in model:
        if name not in shapes:
            report["unexpected"].append(name)
    if verbose:
        for k, v in report.items():
            if v:
                print(f"[checkpoint] {k}: {len(v)} tensors (first: {v[0]})")
    return sd, report


def reset_cls_test(sd: Dict[str, torch.Tensor], zs_weight_path: str, num_classes: int) -> None:
    """Swap the zero-shot classifier at test time (`Detic/detic/modeling/utils.py:32-50`)."""
    w = load_zs_weight(zs_weight_path)
    assert w.shape[1] == num_classes + 1
    for k in range(NUM_CASCADE):
        sd[f"roi_heads.box_predictor.{k}.cls_score.zs_weight"] = w.clone()


def fill_missing(sd: Dict[str, torch.Tensor], seed: int = 0, num_classes: int = 20) -> Dict[str, torch.Tensor]:
    """Complete a partially loaded checkpoint with synthetic tensors (e.g. `map_merge_projection*`
    when starting from a plain Detic checkpoint)."""
    syn = synthetic_state_dict(seed, num_classes)
    out = OrderedDict()
    for name in expected_shapes(num_classes):
        out[name] = sd[name] if name in sd else syn[name]
    return out


# ----------------------------------------------------------------------------------------------------------------------------------
# Checkpoint OUT: the parameters a `modeling.training.Trainer` stepped, back in the reference's names and layouts, written in the
# format `load_checkpoint` (and detectron2's DetectionCheckpointer, train_mp3d.py:523-531,659) reads.
# ----------------------------------------------------------------------------------------------------------------------------------
def unpack_parameter(name: str, packed: torch.Tensor, shape: Tuple[int, ...]) -> torch.Tensor:
    """A stepped tensor in the layout the kernels keep it in -> the reference's tensor of `shape`.
    Convolutions: packed rows [Cout, (ky, kx, c)] (c padded to 4 for the stem) -> OIHW; `box_head.k.fc1`: columns in the pooled rows'
    (7, 7, C) order -> the reference's flatten order (C, 7, 7); everything else is a reshape."""
    t = packed.detach().to(torch.float32).cpu()
    if len(shape) == 4:
        O, I, KH, KW = shape
        ipad = t.shape[1] // (KH * KW)
        if t.shape[0] != O or ipad < I:
            raise ValueError(f"{name}: packed {tuple(t.shape)} does not hold {shape}")
        return t[:, :KH * KW * ipad].reshape(O, KH, KW, ipad)[..., :I].permute(0, 3, 1, 2).contiguous()
    if len(shape) == 2 and name.endswith("fc1.weight") and "box_head" in name:
        O, K = shape
        c = K // 49
        return t[:, :K].reshape(O, 7, 7, c).permute(0, 3, 1, 2).reshape(O, K).contiguous()
    if len(shape) == 2:
        return t[:shape[0], :shape[1]].contiguous()
    return t.reshape(shape).contiguous()


def export_state_dict(entries, base_sd: Dict[str, torch.Tensor], num_classes: int = 20) -> "OrderedDict[str, torch.Tensor]":
    """`entries`: (reference parameter name, stepped tensor, ...) as `Trainer.entries` lists them; `base_sd`: the state dict the model was
    built from (buffers and untrained tensors are taken from it) -> a complete reference-keyed state dict."""
    shapes = expected_shapes(num_classes)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict((k, v.detach().clone() if torch.is_tensor(v) else v) for k, v in base_sd.items())
    for e in entries:
        name, tensor = e[0], e[1]
        if name.endswith("centernet_head.scales"):                 # five `Scale` modules stepped as one tensor
            for l, v in enumerate(tensor.detach().cpu().tolist()):
                out[f"{name}.{l}.scale"] = torch.tensor([v], dtype=torch.float32)
            continue
        if name not in shapes:
            raise KeyError(f"{name}: not a tensor of the reference's state dict")
        out[name] = unpack_parameter(name, tensor, shapes[name])
    return out


def save_checkpoint(path: str, sd: Dict[str, torch.Tensor], iteration: int = 0, optimizer: Optional[Dict] = None,
                    scheduler: Optional[Dict] = None) -> None:
    """What DetectionCheckpointer.save writes (train_mp3d.py:521-523,654): {'model': state dict, 'iteration': n} and, when given, the
    checkpointables 'optimizer' / 'scheduler'; the directory's `last_checkpoint` file names the newest save (fvcore
    `Checkpointer.tag_last_checkpoint`), which is what `--resume` follows."""
    obj = {"model": OrderedDict((k, v) for k, v in sd.items()), "iteration": int(iteration)}
    if optimizer is not None:
        obj["optimizer"] = optimizer
    if scheduler is not None:
        obj["scheduler"] = scheduler
    torch.save(obj, path)
    with open(os.path.join(os.path.dirname(os.path.abspath(path)), "last_checkpoint"), "w") as fh:
        fh.write(os.path.basename(path))


def last_checkpoint(output_dir: str) -> Optional[str]:
    """fvcore `Checkpointer.get_checkpoint_file`: the file `last_checkpoint` names, or None (`resume_or_load` then falls back to
    MODEL.WEIGHTS and the run starts at iteration 0)."""
    tag = os.path.join(output_dir, "last_checkpoint")
    if not os.path.exists(tag):
        return None
    with open(tag) as fh:
        name = fh.read().strip()
    path = os.path.join(output_dir, name)
    return path if os.path.exists(path) else None


def load_training_state(path: str) -> Dict:
    """The non-model part of a checkpoint written by `save_checkpoint`: {'iteration', 'optimizer', 'scheduler'} for
    `engine.train_loop.do_train(resume_state=...)`.  A file without a scheduler entry (weights only) resumes the schedule at
    its stored iteration."""
    obj = torch.load(path, map_location="cpu", weights_only=False)
    it = int(obj.get("iteration", -1)) if isinstance(obj, dict) else -1
    return {"iteration": it, "optimizer": obj.get("optimizer") if isinstance(obj, dict) else None,
            "scheduler": (obj.get("scheduler") if isinstance(obj, dict) else None) or {"last_epoch": max(it, 0)}}
