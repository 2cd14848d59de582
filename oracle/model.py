"""Oracle of the dense per-frame path (a5-a16 of SURVEY.md §8a).  TEST INFRASTRUCTURE ONLY.

torch fp32 on CPU, NCHW, consumes a reference-keyed state dict (plain dict of tensors).  Each
function cites the reference lines it restates.  detectron2/timm-owned pieces follow SURVEY.md
Appendix A (parity unpinned there, see `oracle/__init__.py`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops

PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)
FPN_STRIDES = (8, 16, 32, 64, 128)
CASCADE_WEIGHTS = ((10.0, 10.0, 5.0, 5.0), (20.0, 20.0, 10.0, 10.0), (30.0, 30.0, 15.0, 15.0))


@dataclass
class OracleCfg:
    """The config keys that steer the path, with the values of
    `Detic/configs/Base-C2_L_R5021k_640b64_4x_recurrent.yaml` + `Detic/detic/config.py`."""
    memory_type: str = "implicit_memory"        # MODEL.MEMORY_TYPE
    map_feat_fusion: str = "sum"                 # MODEL.MAP_FEAT_FUSION
    map_feature_weight: float = 5.0              # MODEL.MAP_FEATURE_WEIGHT (README runs use 5)
    memory_cls_score_thresh: float = 0.3         # MODEL.MEMORY_CLS_SCORE_THRESH
    test_type: str = "default"                   # MODEL.TEST_TYPE
    inference_th: float = 0.0001                 # MODEL.CENTERNET.INFERENCE_TH
    pre_nms_topk: int = 1000                     # MODEL.CENTERNET.PRE_NMS_TOPK_TEST
    post_nms_topk: int = 256                     # MODEL.CENTERNET.POST_NMS_TOPK_TEST
    nms_th_proposal: float = 0.9                 # MODEL.CENTERNET.NMS_TH_TEST
    score_thresh_test: float = 0.02              # MODEL.ROI_HEADS.SCORE_THRESH_TEST
    nms_thresh_test: float = 0.5                 # MODEL.ROI_HEADS.NMS_THRESH_TEST
    detections_per_image: int = 300              # TEST.DETECTIONS_PER_IMAGE
    num_classes: int = 20
    norm_temp: float = 50.0                      # MODEL.ROI_BOX_HEAD.NORM_TEMP
    mask_threshold: float = 0.5
    pixel_mean: Tuple[float, float, float] = PIXEL_MEAN
    pixel_std: Tuple[float, float, float] = PIXEL_STD


# ----------------------------------------------------------------------------------------------
# a5  preprocess_image (d2 GeneralizedRCNN.preprocess_image, called `custom_rcnn.py:557`) - A1
# ----------------------------------------------------------------------------------------------
def preprocess_image(image_u8: torch.Tensor, cfg: OracleCfg, div: int = 32) -> torch.Tensor:
    """u8 [3,H,W] RGB -> f32 [1,3,H',W'] normalised, zero padded (after normalisation) to /div."""
    x = image_u8.float()
    mean = torch.tensor(cfg.pixel_mean, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(cfg.pixel_std, dtype=torch.float32).view(3, 1, 1)
    x = (x - mean) / std
    H, W = x.shape[1:]
    Hp = (H + div - 1) // div * div
    Wp = (W + div - 1) // div * div
    out = torch.zeros((1, 3, Hp, Wp), dtype=torch.float32)
    out[0, :, :H, :W] = x
    return out


# ----------------------------------------------------------------------------------------------
# a6  timm-0.5.4 ResNet-50 with FrozenBatchNorm2d (`timm.py:277-299`, A2, A3)
# ----------------------------------------------------------------------------------------------
def frozen_bn(x: torch.Tensor, sd, prefix: str, eps: float = 1e-5) -> torch.Tensor:
    return F.batch_norm(x, sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"],
                        sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], training=False, eps=eps)


def bottleneck(x: torch.Tensor, sd, p: str, stride: int) -> torch.Tensor:
    shortcut = x
    out = F.relu(frozen_bn(F.conv2d(x, sd[f"{p}.conv1.weight"]), sd, f"{p}.bn1"))
    out = F.relu(frozen_bn(F.conv2d(out, sd[f"{p}.conv2.weight"], stride=stride, padding=1), sd, f"{p}.bn2"))
    out = frozen_bn(F.conv2d(out, sd[f"{p}.conv3.weight"]), sd, f"{p}.bn3")
    if f"{p}.downsample.0.weight" in sd:
        shortcut = frozen_bn(F.conv2d(x, sd[f"{p}.downsample.0.weight"], stride=stride), sd, f"{p}.downsample.1")
    return F.relu(out + shortcut)


def resnet50(x: torch.Tensor, sd) -> Dict[str, torch.Tensor]:
    base = "backbone.bottom_up.base"
    x = F.relu(frozen_bn(F.conv2d(x, sd[f"{base}.conv1.weight"], stride=2, padding=3), sd, f"{base}.bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = {}
    for li, nblk in enumerate((3, 4, 6, 3), start=1):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 1) else 1
            x = bottleneck(x, sd, f"{base}.layer{li}.{b}", stride)
        feats[f"layer{li + 1}"] = x  # layer2 output is exposed as 'layer3' etc. (`timm.py:379,404`)
    return feats


# ----------------------------------------------------------------------------------------------
# a7-a9  CustomRecurrentFPN.forward (`timm.py:91-213`) incl. memory read + fusion (142-192)
# ----------------------------------------------------------------------------------------------
def fpn_top_down(c: Dict[str, torch.Tensor], sd) -> List[torch.Tensor]:
    """`timm.py:118-136`: lateral 1x1, nearest x2 + add, 3x3 output.  Returns [P3,P4,P5]."""
    def lat(l, x):
        return F.conv2d(x, sd[f"backbone.fpn_lateral{l}.weight"], sd[f"backbone.fpn_lateral{l}.bias"])

    def outc(l, x):
        return F.conv2d(x, sd[f"backbone.fpn_output{l}.weight"], sd[f"backbone.fpn_output{l}.bias"], padding=1)

    prev = lat(5, c["layer5"])
    results = [outc(5, prev)]
    for l, name in ((4, "layer4"), (3, "layer3")):
        top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = lat(l, c[name]) + top_down
        results.insert(0, outc(l, prev))
    return results


def memory_read_pooled(memory_f16: torch.Tensor, proj: torch.Tensor) -> List[torch.Tensor]:
    """`timm.py:147-168`: gather fp16 memory rows by proj index, avg_pool 4 (f32), then per level
    avg_pool 2 (f32) -> fp16, cascaded.  Returns the three fp16 tensors [1,512,h,w] (P3,P4,P5)."""
    ego = memory_f16[proj].permute(2, 0, 1).unsqueeze(0)           # [1,512,H,W] fp16
    ego = F.avg_pool2d(ego.to(torch.float32), kernel_size=4, stride=4)
    out = []
    cur = ego
    for _ in range(3):
        cur = F.avg_pool2d(cur.to(torch.float32), kernel_size=2, stride=2).to(torch.half)
        out.append(cur)
    return out


def fuse_memory(results: List[torch.Tensor], pooled: List[torch.Tensor], sd, cfg: OracleCfg) -> List[torch.Tensor]:
    """`timm.py:163-192`: 1x1 projection (f32) x MAP_FEATURE_WEIGHT, fusion."""
    new = []
    for i, res in enumerate(results):
        mem = F.conv2d(pooled[i].to(torch.float32), sd[f"backbone.map_merge_projection{i + 1}.weight"],
                       sd[f"backbone.map_merge_projection{i + 1}.bias"])
        mem = mem * cfg.map_feature_weight
        if cfg.map_feat_fusion == "sum":
            r = mem + res
        elif cfg.map_feat_fusion == "mem_only":
            r = mem
        elif cfg.map_feat_fusion == "image_only":
            r = res
        else:
            raise ValueError(cfg.map_feat_fusion)
        new.append(r.to(res.dtype))
    return new


def top_block(p5: torch.Tensor, sd) -> List[torch.Tensor]:
    """`LastLevelP6P7_P5.forward` `timm.py:359-364`."""
    p6 = F.conv2d(p5, sd["backbone.top_block.p6.weight"], sd["backbone.top_block.p6.bias"], stride=2, padding=1)
    p7 = F.conv2d(F.relu(p6), sd["backbone.top_block.p7.weight"], sd["backbone.top_block.p7.bias"], stride=2, padding=1)
    return [p6, p7]


def backbone_forward(x: torch.Tensor, sd, cfg: OracleCfg, memory_f16: Optional[torch.Tensor],
                     proj: Optional[torch.Tensor]) -> List[torch.Tensor]:
    """Returns [P3..P7] (each [1,256,h,w])."""
    c = resnet50(x, sd)
    results = fpn_top_down(c, sd)
    if cfg.memory_type == "implicit_memory":
        pooled = memory_read_pooled(memory_f16, proj)
        results = fuse_memory(results, pooled, sd, cfg)
    results.extend(top_block(results[2], sd))
    return results


# ----------------------------------------------------------------------------------------------
# a10  CenterNetHead.forward (`centernet_head.py:141-161`), ONLY_PROPOSAL + WITH_AGN_HM
# ----------------------------------------------------------------------------------------------
def centernet_head(feats: List[torch.Tensor], sd) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    h = "proposal_generator.centernet_head"
    agn, reg = [], []
    for l, f in enumerate(feats):
        t = f
        for i in range(4):
            t = F.conv2d(t, sd[f"{h}.bbox_tower.{3 * i}.weight"], sd[f"{h}.bbox_tower.{3 * i}.bias"], padding=1)
            t = F.group_norm(t, 32, sd[f"{h}.bbox_tower.{3 * i + 1}.weight"], sd[f"{h}.bbox_tower.{3 * i + 1}.bias"], eps=1e-5)
            t = F.relu(t)
        agn.append(F.conv2d(t, sd[f"{h}.agn_hm.weight"], sd[f"{h}.agn_hm.bias"], padding=1))
        r = F.conv2d(t, sd[f"{h}.bbox_pred.weight"], sd[f"{h}.bbox_pred.bias"], padding=1)
        r = r * sd[f"{h}.scales.{l}.scale"]
        reg.append(F.relu(r))
    return agn, reg


# ----------------------------------------------------------------------------------------------
# a11  CenterNet.inference / predict_instances / predict_single_level / nms_and_topK
#      (`centernet.py:603-745`), compute_grids (321-339), ml_nms (`layers/ml_nms.py:4-31`)
# ----------------------------------------------------------------------------------------------
def centernet_proposals(agn: List[torch.Tensor], reg: List[torch.Tensor], cfg: OracleCfg):
    """Returns (boxes [R,4], scores [R]) sorted by score descending.

    Tie rules chosen by this build where upstream is implementation-defined: the unsorted per-level
    top-k keeps, among equal scores at the cut, the lower flat index; NMS order among equal scores is
    lower concatenated index first."""
    all_boxes, all_scores = [], []
    for l, (a, r) in enumerate(zip(agn, reg)):
        stride = FPN_STRIDES[l]
        _, _, H, W = a.shape
        heat = torch.sigmoid(a)[0, 0].reshape(-1)                        # HW (C == 1)
        regl = (r * stride)[0].permute(1, 2, 0).reshape(-1, 4)            # HW x 4
        sx = torch.arange(0, W * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(0, H * stride, step=stride, dtype=torch.float32)
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        grids = torch.stack((gx.reshape(-1), gy.reshape(-1)), dim=1) + stride // 2
        cand = torch.nonzero(heat > cfg.inference_th).squeeze(1)
        sc = heat[cand]
        if cand.numel() > cfg.pre_nms_topk:
            order = torch.sort(sc, descending=True, stable=True).indices[:cfg.pre_nms_topk]
            order = torch.sort(order).values                     # keep flat-index order
            cand, sc = cand[order], sc[order]
        rg, gd = regl[cand], grids[cand]
        det = torch.stack([gd[:, 0] - rg[:, 0], gd[:, 1] - rg[:, 1], gd[:, 0] + rg[:, 2], gd[:, 1] + rg[:, 3]], dim=1)
        det[:, 2] = torch.max(det[:, 2], det[:, 0] + 0.01)
        det[:, 3] = torch.max(det[:, 3], det[:, 1] + 0.01)
        all_boxes.append(det)
        all_scores.append(torch.sqrt(sc))
    boxes = torch.cat(all_boxes)
    scores = torch.cat(all_scores)
    keep = ops.nms(boxes, scores, cfg.nms_th_proposal)   # class agnostic (labels all zero)
    boxes, scores = boxes[keep], scores[keep]
    n = boxes.shape[0]
    if n > cfg.post_nms_topk:
        kth = torch.kthvalue(scores, n - cfg.post_nms_topk + 1).values
        k = torch.nonzero(scores >= kth).squeeze(1)
        boxes, scores = boxes[k], scores[k]
    return boxes, scores


# ----------------------------------------------------------------------------------------------
# a12-a13  DeticCascadeROIHeads._forward_box (`detic_roi_heads.py:88-222,306-349`)
# ----------------------------------------------------------------------------------------------
def box_head_stage(pooled: torch.Tensor, sd, k: int, cfg: OracleCfg):
    """FastRCNNConvFCHead (A8) + DeticFastRCNNOutputLayers.forward (`detic_fast_rcnn.py:437-466`)
    + ZeroShotClassifier.forward (`zero_shot_classifier.py:71-111`).
    Returns (logits [R,C+1], deltas [R,4], clip_feat [R,512])."""
    x = pooled.flatten(1)
    x = F.relu(F.linear(x, sd[f"roi_heads.box_head.{k}.fc1.weight"], sd[f"roi_heads.box_head.{k}.fc1.bias"]))
    x = F.relu(F.linear(x, sd[f"roi_heads.box_head.{k}.fc2.weight"], sd[f"roi_heads.box_head.{k}.fc2.bias"]))
    p = f"roi_heads.box_predictor.{k}"
    feat = F.linear(x, sd[f"{p}.cls_score.linear.weight"], sd[f"{p}.cls_score.linear.bias"])
    xn = cfg.norm_temp * F.normalize(feat, p=2, dim=1)
    logits = torch.mm(xn, sd[f"{p}.cls_score.zs_weight"])
    d = F.relu(F.linear(x, sd[f"{p}.bbox_pred.0.weight"], sd[f"{p}.bbox_pred.0.bias"]))
    deltas = F.linear(d, sd[f"{p}.bbox_pred.2.weight"], sd[f"{p}.bbox_pred.2.bias"])
    return logits, deltas, feat


def cascade_box_heads(feats: List[torch.Tensor], prop_boxes: torch.Tensor, prop_scores: torch.Tensor,
                      sd, cfg: OracleCfg, image_hw: Tuple[int, int]):
    """Returns dict(final_boxes [R,4], final_scores [R,C+1], feat0 [R,512], stage_boxes)."""
    p345 = feats[:3]
    boxes = prop_boxes
    stage_probs = []
    feat0 = None
    stage_boxes = []
    for k in range(3):
        if k > 0:
            boxes = ops.clip_boxes(boxes, image_hw)     # `_create_proposals_from_boxes` :314
        stage_boxes.append(boxes)
        pooled = ops.roi_pool(p345, boxes, 7)
        logits, deltas, feat = box_head_stage(pooled, sd, k, cfg)
        if k == 0:
            feat0 = feat                                 # `_run_stage` :339-346 (ADD_FEATURE_TO_PROP)
        stage_probs.append(torch.sigmoid(logits))        # predict_probs, USE_SIGMOID_CE
        boxes = ops.apply_deltas(deltas, boxes, CASCADE_WEIGHTS[k])
    scores = (stage_probs[0] + stage_probs[1] + stage_probs[2]) * (1.0 / 3)
    scores = (scores * prop_scores[:, None]) ** 0.5      # MULT_PROPOSAL_SCORE :171-173
    return dict(final_boxes=boxes, final_scores=scores, feat0=feat0, stage_boxes=stage_boxes)


# ----------------------------------------------------------------------------------------------
# a14  mask head (A11): MaskRCNNConvUpsampleHead.layers + mask_rcnn_inference
# ----------------------------------------------------------------------------------------------
def mask_head(feats: List[torch.Tensor], boxes: torch.Tensor, sd, chunk: int = 64) -> torch.Tensor:
    """-> mask probabilities [R,1,28,28] (class-agnostic, sigmoid)."""
    m = "roi_heads.mask_head"
    outs = []
    for s in range(0, boxes.shape[0], chunk):
        x = ops.roi_pool(feats[:3], boxes[s:s + chunk], 14)
        for i in range(1, 5):
            x = F.relu(F.conv2d(x, sd[f"{m}.mask_fcn{i}.weight"], sd[f"{m}.mask_fcn{i}.bias"], padding=1))
        x = F.relu(F.conv_transpose2d(x, sd[f"{m}.deconv.weight"], sd[f"{m}.deconv.bias"], stride=2))
        x = F.conv2d(x, sd[f"{m}.predictor.weight"], sd[f"{m}.predictor.bias"])
        outs.append(torch.sigmoid(x))
    if not outs:
        return torch.zeros((0, 1, 28, 28), dtype=torch.float32)
    return torch.cat(outs)


# ----------------------------------------------------------------------------------------------
# a15  detector_postprocess (A13; `custom_rcnn.py:579-580`)
# ----------------------------------------------------------------------------------------------
def detector_postprocess(boxes, scores, classes, masks28, image_hw, out_hw, cfg: OracleCfg):
    sx = out_hw[1] / image_hw[1]
    sy = out_hw[0] / image_hw[0]
    b = boxes.clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b = ops.clip_boxes(b, out_hw)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    b, scores, classes, masks28 = b[keep], scores[keep], classes[keep], masks28[keep]
    pm = ops.paste_masks(masks28[:, 0], b, out_hw, cfg.mask_threshold)
    return dict(pred_boxes=b, scores=scores, pred_classes=classes, pred_masks=pm)


# ----------------------------------------------------------------------------------------------
# `CustomRCNNRecurrent.inference` (`custom_rcnn.py:548-582`)
# ----------------------------------------------------------------------------------------------
def inference(sd, cfg: OracleCfg, image_u8: torch.Tensor, memory_f16: Optional[torch.Tensor],
              proj: Optional[torch.Tensor], out_hw: Optional[Tuple[int, int]] = None, want_intermediates: bool = False,
              timings: Optional[dict] = None):
    """`timings` (optional dict): seconds per stage are ADDED to it (bench.py's cpu_baseline leg: SURVEY 8d per-stage breakdown)."""
    import time as _t
    t_last = [_t.perf_counter()]

    def lap(name):
        if timings is not None:
            now = _t.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - t_last[0])
            t_last[0] = now

    H, W = image_u8.shape[1:]
    image_hw = (H, W)
    x = preprocess_image(image_u8, cfg)
    feats = backbone_forward(x, sd, cfg, memory_f16, proj)
    lap("preprocess + backbone + FPN + memory read/fusion")
    agn, reg = centernet_head(feats, sd)
    prop_boxes, prop_scores = centernet_proposals(agn, reg, cfg)
    lap("CenterNet head + proposal decoding")
    cas = cascade_box_heads(feats, prop_boxes, prop_scores, sd, cfg, image_hw)
    det_boxes, det_scores, det_classes, _ = ops.fast_rcnn_inference_single(
        cas["final_boxes"], cas["final_scores"], image_hw, cfg.score_thresh_test, cfg.nms_thresh_test,
        cfg.detections_per_image)
    lap("cascade box heads + fast_rcnn_inference")
    det_masks = mask_head(feats, det_boxes, sd)                 # forward_with_given_boxes
    lap("mask head (detections)")
    prop_masks = mask_head(feats, prop_boxes, sd)               # forward_mask_memory `:573-574`
    lap("mask head (proposals)")
    result = detector_postprocess(det_boxes, det_scores, det_classes, det_masks, image_hw, out_hw or image_hw, cfg)
    lap("detector_postprocess + paste")
    proposals = dict(proposal_boxes=prop_boxes, scores=prop_scores, feat=cas["feat0"], pred_masks=prop_masks)
    if want_intermediates:
        inter = dict(feats=feats, agn=agn, reg=reg, cascade=cas, det_boxes=det_boxes, det_scores=det_scores,
                     det_classes=det_classes, det_masks=det_masks, x=x)
        return proposals, result, inter
    return proposals, result
