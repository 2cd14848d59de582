"""Oracle restatements of the detectron2 / torchvision ops the hot path calls.  TEST INFRASTRUCTURE.

Upstream packages are not in the reference tree: parity unpinned; semantics follow
SURVEY.md Appendix A (A6, A7, A9, A10, A12, A13).  All tensors fp32 CPU, layouts NCHW as upstream.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# NMS (A6; call sites `Detic/third_party/CenterNet2/centernet/modeling/layers/ml_nms.py:27`
# and detectron2 `fast_rcnn_inference`)
# --------------------------------------------------------------------------------------------
def box_area(b: torch.Tensor) -> torch.Tensor:
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def iou_one_to_many(b: torch.Tensor, bs: torch.Tensor) -> torch.Tensor:
    lt = torch.maximum(b[:2], bs[:, :2])
    rb = torch.minimum(b[2:], bs[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    a = (b[2] - b[0]) * (b[3] - b[1])
    return inter / (a + box_area(bs) - inter)


def nms(boxes: torch.Tensor, scores: torch.Tensor, thr: float) -> torch.Tensor:
    """Greedy NMS.  Sort by score descending, ties -> lower original index first (stable);
    suppress when IoU > thr (strict).  Returns kept indices in descending score order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order].float()
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            iou = iou_one_to_many(b[i], b[i + 1:])
            suppressed[i + 1:] |= iou > thr
    return order[torch.tensor(keep, dtype=torch.int64)]


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, thr: float) -> torch.Tensor:
    """Per-class NMS on the original coordinates (the `_batched_nms_vanilla` semantics of torchvision;
    the coordinate-offset trick variant is the same algorithm up to fp32 rounding of the shifted
    coordinates, which is NOT reproduced here).  Returns kept indices sorted by score descending
    (stable: equal scores keep ascending index order)."""
    if boxes.numel() == 0:
        return torch.zeros((0,), dtype=torch.int64)
    keep_mask = torch.zeros(boxes.shape[0], dtype=torch.bool)
    for c in torch.unique(idxs):
        ci = torch.nonzero(idxs == c).squeeze(1)
        k = nms(boxes[ci], scores[ci], thr)
        keep_mask[ci[k]] = True
    kept = torch.nonzero(keep_mask).squeeze(1)
    order = torch.sort(scores[kept], descending=True, stable=True).indices
    return kept[order]


# --------------------------------------------------------------------------------------------
# Box2BoxTransform.apply_deltas (A9; via `predict_boxes`,
# `Detic/detic/modeling/roi_heads/detic_roi_heads.py:121-122,179-180`)
# --------------------------------------------------------------------------------------------
SCALE_CLAMP = math.log(1000.0 / 16)


def apply_deltas(deltas: torch.Tensor, boxes: torch.Tensor, weights: Tuple[float, float, float, float]) -> torch.Tensor:
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0] / wx
    dy = deltas[:, 1] / wy
    dw = torch.clamp(deltas[:, 2] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[:, 3] / wh, max=SCALE_CLAMP)
    pcx = dx * widths + ctr_x
    pcy = dy * heights + ctr_y
    pw = torch.exp(dw) * widths
    ph = torch.exp(dh) * heights
    return torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], dim=1)


def clip_boxes(boxes: torch.Tensor, hw: Tuple[int, int]) -> torch.Tensor:
    h, w = hw
    x1 = boxes[:, 0].clamp(min=0, max=w)
    y1 = boxes[:, 1].clamp(min=0, max=h)
    x2 = boxes[:, 2].clamp(min=0, max=w)
    y2 = boxes[:, 3].clamp(min=0, max=h)
    return torch.stack([x1, y1, x2, y2], dim=1)


# --------------------------------------------------------------------------------------------
# ROIPooler + ROIAlign(aligned=True) (A7; box pooler `detic_roi_heads.py:332`, mask pooler `:265`)
# --------------------------------------------------------------------------------------------
def assign_boxes_to_levels(boxes: torch.Tensor, min_level: int = 3, max_level: int = 5,
                           canonical_box_size: int = 224, canonical_level: int = 4) -> torch.Tensor:
    box_sizes = torch.sqrt(box_area(boxes))
    lvl = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + 1e-8))
    lvl = torch.clamp(lvl, min=min_level, max=max_level)
    return lvl.to(torch.int64) - min_level


def _bilinear(feat: torch.Tensor, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """feat [C,H,W]; y,x [S] sample coords -> [C,S] (torchvision roi_align bilinear_interpolate)."""
    C, H, W = feat.shape
    invalid = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = y.clamp(min=0)
    x = x.clamp(min=0)
    y_low = y.to(torch.int64)
    x_low = x.to(torch.int64)
    yc = y_low >= H - 1
    xc = x_low >= W - 1
    y_high = torch.where(yc, torch.full_like(y_low, H - 1), y_low + 1)
    y_low = torch.where(yc, torch.full_like(y_low, H - 1), y_low)
    y = torch.where(yc, y_low.to(y.dtype), y)
    x_high = torch.where(xc, torch.full_like(x_low, W - 1), x_low + 1)
    x_low = torch.where(xc, torch.full_like(x_low, W - 1), x_low)
    x = torch.where(xc, x_low.to(x.dtype), x)
    ly = y - y_low.to(y.dtype)
    lx = x - x_low.to(x.dtype)
    hy = 1.0 - ly
    hx = 1.0 - lx
    w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
    v1 = feat[:, y_low, x_low]
    v2 = feat[:, y_low, x_high]
    v3 = feat[:, y_high, x_low]
    v4 = feat[:, y_high, x_high]
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    return torch.where(invalid[None, :], torch.zeros_like(val), val)


def roi_align_single(feat: torch.Tensor, box: torch.Tensor, scale: float, out: int) -> torch.Tensor:
    """feat [C,H,W], one box (x1,y1,x2,y2) -> [C,out,out]; aligned=True, sampling_ratio=0."""
    C = feat.shape[0]
    # fp32 arithmetic as the upstream kernel (T = float)
    box_s = box.float() * scale - 0.5
    x1, y1, x2, y2 = box_s[0], box_s[1], box_s[2], box_s[3]
    roi_w = x2 - x1
    roi_h = y2 - y1
    bin_h = roi_h / out
    bin_w = roi_w / out
    # ceil on the fp32 quotient, as upstream: ceil(roi_height / pooled_height) in T
    gh = int(torch.ceil(roi_h / out).item())
    gw = int(torch.ceil(roi_w / out).item())
    count = max(gh * gw, 1)
    if gh <= 0 or gw <= 0:
        return torch.zeros((C, out, out), dtype=torch.float32)
    ph = torch.arange(out, dtype=torch.float32)
    iy = torch.arange(gh, dtype=torch.float32)
    ix = torch.arange(gw, dtype=torch.float32)
    ys = y1 + ph[:, None] * bin_h + (iy[None, :] + 0.5) * bin_h / gh  # [out, gh]
    xs = x1 + ph[:, None] * bin_w + (ix[None, :] + 0.5) * bin_w / gw  # [out, gw]
    Y = ys[:, None, :, None].expand(out, out, gh, gw).reshape(-1)
    X = xs[None, :, None, :].expand(out, out, gh, gw).reshape(-1)
    v = _bilinear(feat, Y, X).reshape(C, out, out, gh * gw)
    # upstream accumulates samples sequentially (iy outer, ix inner) then divides by count
    acc = torch.zeros((C, out, out), dtype=torch.float32)
    for s in range(gh * gw):
        acc = acc + v[..., s]
    return acc / count


def roi_pool(feats: List[torch.Tensor], boxes: torch.Tensor, out: int,
             scales=(1.0 / 8, 1.0 / 16, 1.0 / 32)) -> torch.Tensor:
    """ROIPooler over p3..p5 (feats: list of [1,C,H,W]); rows in input-box order."""
    R = boxes.shape[0]
    C = feats[0].shape[1]
    res = torch.zeros((R, C, out, out), dtype=torch.float32)
    if R == 0:
        return res
    lv = assign_boxes_to_levels(boxes)
    for r in range(R):
        l = int(lv[r])
        res[r] = roi_align_single(feats[l][0], boxes[r], scales[l], out)
    return res


# --------------------------------------------------------------------------------------------
# fast_rcnn_inference single image (A10; calls `detic_roi_heads.py:214`, `custom_rcnn.py:862`)
# --------------------------------------------------------------------------------------------
def fast_rcnn_inference_single(boxes: torch.Tensor, scores: torch.Tensor, image_hw: Tuple[int, int],
                               score_thresh: float, nms_thresh: float, topk: int):
    """boxes [R,4] class-agnostic, scores [R,C+1].  Returns (boxes, scores, classes, row_index)."""
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    rows = torch.arange(boxes.shape[0])
    if not bool(valid.all()):
        boxes, scores, rows = boxes[valid], scores[valid], rows[valid]
    scores = scores[:, :-1]
    boxes = clip_boxes(boxes, image_hw)
    mask = scores > score_thresh
    inds = torch.nonzero(mask)  # row-major (row, class)
    cand_boxes = boxes[inds[:, 0]]
    cand_scores = scores[mask]
    keep = batched_nms(cand_boxes, cand_scores, inds[:, 1], nms_thresh)
    if topk >= 0:
        keep = keep[:topk]
    return cand_boxes[keep], cand_scores[keep], inds[keep, 1], rows[inds[keep, 0]]


# --------------------------------------------------------------------------------------------
# paste_masks_in_image (A12; calls `custom_rcnn.py:880` and inside `detector_postprocess`)
# --------------------------------------------------------------------------------------------
def paste_masks_prob(masks: torch.Tensor, boxes: torch.Tensor, hw: Tuple[int, int], chunk: int = 16) -> torch.Tensor:
    """masks [K,28,28] probabilities, boxes [K,4] -> f32 [K,H,W]: the bilinear samples `paste_masks` thresholds.

    Full-image grid_sample form of detectron2 `_do_paste_mask(skip_empty=False)`."""
    K = masks.shape[0]
    H, W = hw
    out = torch.zeros((K, H, W), dtype=torch.float32)
    for s in range(0, K, chunk):
        m = masks[s:s + chunk, None].float()
        b = boxes[s:s + chunk].float()
        x0, y0, x1, y1 = b[:, 0:1], b[:, 1:2], b[:, 2:3], b[:, 3:4]
        img_y = torch.arange(0, H, dtype=torch.float32) + 0.5
        img_x = torch.arange(0, W, dtype=torch.float32) + 0.5
        img_y = (img_y - y0) / (y1 - y0) * 2 - 1
        img_x = (img_x - x0) / (x1 - x0) * 2 - 1
        n = m.shape[0]
        gx = img_x[:, None, :].expand(n, H, W)
        gy = img_y[:, :, None].expand(n, H, W)
        grid = torch.stack([gx, gy], dim=3)
        out[s:s + chunk] = F.grid_sample(m, grid, align_corners=False)[:, 0]
    return out


def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, hw: Tuple[int, int], threshold: float = 0.5,
                chunk: int = 16) -> torch.Tensor:
    """masks [K,28,28] probabilities, boxes [K,4] -> bool [K,H,W] (`paste_masks_prob` >= threshold)."""
    K = masks.shape[0]
    H, W = hw
    out = torch.zeros((K, H, W), dtype=torch.bool)
    for s in range(0, K, chunk):
        out[s:s + chunk] = paste_masks_prob(masks[s:s + chunk], boxes[s:s + chunk], hw, chunk) >= threshold
    return out
