"""CPU restatement of the proposal generator's training losses  --  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline).

CenterNet `losses()` for the recurrent configuration (ONLY_PROPOSAL + WITH_AGN_HM: class-agnostic heatmap focal loss + GIoU
regression loss; `Detic/third_party/CenterNet2/centernet/modeling/dense_heads/centernet.py:241-318`), with the two functions it calls:
`binary_heatmap_focal_loss` (`.../layers/heatmap_focal_loss.py:52-84`) and `IOULoss.forward` (`.../layers/iou_loss.py:10-64`).
Plain differentiable torch on CPU tensors; pinned by `tests/golden/centernet_loss.npz` (the reference's own functions run on seeded
inputs, `tests/golden/gen_golden_losses.py`).  Single process: `num_pos_avg` / `reg_norm` are what `reduce_sum(...) / num_gpus`
gives on one rank (centernet.py:259-265, 290-293).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def binary_heatmap_focal_loss(logits: torch.Tensor, targets: torch.Tensor, pos_inds: torch.Tensor, alpha: float = -1.0,
                              beta: float = 4.0, gamma: float = 2.0, sigmoid_clamp: float = 1e-4, ignore_high_fp: float = -1.0):
    """heatmap_focal_loss.py:52-84 (without the in-place sigmoid): logits, targets [M]; pos_inds [N] -> (pos_loss, neg_loss)."""
    pred = torch.clamp(torch.sigmoid(logits), min=sigmoid_clamp, max=1 - sigmoid_clamp)
    neg_weights = torch.pow(1 - targets, beta)
    pos_pred = pred[pos_inds]
    pos_loss = torch.log(pos_pred) * torch.pow(1 - pos_pred, gamma)
    neg_loss = torch.log(1 - pred) * torch.pow(pred, gamma) * neg_weights
    if ignore_high_fp > 0:
        neg_loss = (pred < ignore_high_fp).float() * neg_loss
    pos_loss = -pos_loss.sum()
    neg_loss = -neg_loss.sum()
    if alpha >= 0:
        pos_loss = alpha * pos_loss
        neg_loss = (1 - alpha) * neg_loss
    return pos_loss, neg_loss


def giou_ltrb_loss(pred: torch.Tensor, target: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """iou_loss.py:10-64 with loc_loss_type 'giou', reduction 'sum': pred / target [K,4] distances (left, top, right, bottom)."""
    pl, pt, pr, pb = pred.unbind(1)
    tl, tt, tr, tb = target.unbind(1)
    t_area = (tl + tr) * (tt + tb)
    p_area = (pl + pr) * (pt + pb)
    w_i = torch.min(pl, tl) + torch.min(pr, tr)
    h_i = torch.min(pb, tb) + torch.min(pt, tt)
    g_w = torch.max(pl, tl) + torch.max(pr, tr)
    g_h = torch.max(pb, tb) + torch.max(pt, tt)
    ac = g_w * g_h
    a_i = w_i * h_i
    a_u = t_area + p_area - a_i
    ious = (a_i + 1.0) / (a_u + 1.0)
    gious = ious - (ac - a_u) / ac
    return ((1 - gious) * weight).sum()


def centernet_proposal_losses(agn_logits: torch.Tensor, reg_pred: torch.Tensor, agn_heatmap: torch.Tensor, reg_targets: torch.Tensor,
                              pos_inds: torch.Tensor, *, alpha: float = 0.25, beta: float = 4.0, gamma: float = 2.0,
                              sigmoid_clamp: float = 1e-4, ignore_high_fp: float = 0.85, pos_weight: float = 0.5,
                              neg_weight: float = 0.5, reg_weight: float = 1.0) -> Dict[str, torch.Tensor]:
    """centernet.py:241-318 for only_proposal + with_agn_hm + not_norm_reg on one rank.  agn_logits [M], reg_pred [M,4] (after
    scale + ReLU), agn_heatmap [M] (= flattened_hms.max(dim=1)), reg_targets [M,4] (-INF rows where no object), pos_inds [N]."""
    num_pos_avg = max(float(pos_inds.numel()), 1.0)
    reg_inds = torch.nonzero(reg_targets.max(dim=1)[0] >= 0).squeeze(1)
    weight = torch.ones((reg_inds.numel(),), dtype=torch.float32)                 # not_norm_reg (centernet.py:288-289)
    reg_norm = max(float(weight.sum()), 1.0)
    loc = reg_weight * giou_ltrb_loss(reg_pred[reg_inds], reg_targets[reg_inds], weight) / reg_norm
    pos, neg = binary_heatmap_focal_loss(agn_logits.float(), agn_heatmap.float(), pos_inds, alpha, beta, gamma, sigmoid_clamp,
                                         ignore_high_fp)
    return {"loss_centernet_loc": loc, "loss_centernet_agn_pos": pos_weight * pos / num_pos_avg,
            "loss_centernet_agn_neg": neg_weight * neg / num_pos_avg}


# ----------------------------------------------------------------------------------------------------------------------------------
# DeticFastRCNNOutputLayers.losses (`Detic/detic/modeling/roi_heads/detic_fast_rcnn.py:157-197`) for USE_SIGMOID_CE + class-agnostic
# box regression: sigmoid_cross_entropy_loss (:200-233) and box_reg_loss (:270-303, smooth_l1 branch).  The three upstream pieces it
# calls are not in the reference tree and are restated from their published definitions (parity "unpinned" for them, as SURVEY
# Appendix A says of detectron2 / fvcore): Box2BoxTransform.get_deltas, fvcore smooth_l1_loss, nonzero_tuple.
# ----------------------------------------------------------------------------------------------------------------------------------
def get_deltas(src: torch.Tensor, dst: torch.Tensor, weights) -> torch.Tensor:
    """detectron2 Box2BoxTransform.get_deltas: (dx, dy, dw, dh) that take `src` boxes to `dst`."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = dst[:, 2] - dst[:, 0], dst[:, 3] - dst[:, 1]
    tx, ty = dst[:, 0] + 0.5 * tw, dst[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)


def smooth_l1_sum(x: torch.Tensor, y: torch.Tensor, beta: float) -> torch.Tensor:
    """fvcore.nn.smooth_l1_loss(reduction='sum'): plain L1 below beta 1e-5."""
    d = torch.abs(x - y)
    if beta < 1e-5:
        return d.sum()
    return torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta).sum()


def sigmoid_cross_entropy_loss(logits: torch.Tensor, gt_classes: torch.Tensor, class_weight=None) -> torch.Tensor:
    """detic_fast_rcnn.py:200-233: logits [B, C+1] (last column: background, unused), gt_classes [B] in [0, C] (C = background);
    `class_weight` [C]: the product of the federated-loss mask (:213-222) and the zero-frequency mask (:223-225), or None."""
    B, C = logits.shape[0], logits.shape[1] - 1
    target = logits.new_zeros(B, C + 1)
    target[torch.arange(B), gt_classes] = 1
    cls_loss = torch.nn.functional.binary_cross_entropy_with_logits(logits[:, :-1], target[:, :C], reduction="none")
    if class_weight is not None:
        cls_loss = cls_loss * class_weight.view(1, C)
    return cls_loss.sum() / B


def box_reg_loss(proposal_boxes, gt_boxes, pred_deltas, gt_classes, num_classes: int, weights, beta: float = 0.0) -> torch.Tensor:
    """detic_fast_rcnn.py:270-303, class-agnostic deltas, smooth_l1: foreground rows only, normalised by ALL rows."""
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < num_classes)).squeeze(1)
    tgt = get_deltas(proposal_boxes[fg], gt_boxes[fg], weights)
    return smooth_l1_sum(pred_deltas[fg], tgt, beta) / max(gt_classes.numel(), 1.0)


# ----------------------------------------------------------------------------------------------------------------------------------
# CenterNet target assignment for ONLY_PROPOSAL, one image (centernet.py:342-479): `_get_ground_truth` with `compute_grids` (:321-339),
# `_get_label_inds` (:441-479), `assign_fpn_level` (:482-498), `assign_reg_fpn` (:501-513), `_get_reg_targets` (:516-528),
# `_create_agn_heatmaps_from_dist` (:549-560), `get_center3x3` (:577-594).  Pinned by tests/golden/centernet_targets.npz.
# ----------------------------------------------------------------------------------------------------------------------------------
CENTERNET_INF = 100000000          # centernet.py:28
CENTERNET_SOI = ((0, 80), (64, 160), (128, 320), (256, 640), (512, 10000000))


def centernet_targets(gt_boxes: torch.Tensor, level_hw, strides=(8, 16, 32, 64, 128), sizes_of_interest=CENTERNET_SOI,
                      hm_min_overlap: float = 0.8, min_radius: float = 4.0):
    """gt_boxes [N,4] (x1,y1,x2,y2), level_hw [(h, w)] per level -> (pos_inds [N'] int64, reg_targets [M,4] in units of the level's
    stride with -INF / stride rows where no object claims the position, agn_heatmap [M]), M = sum h*w in level order."""
    delta = (1 - hm_min_overlap) / (1 + hm_min_overlap)
    M = sum(h * w for h, w in level_hw)
    N = gt_boxes.shape[0]
    regs, heats, pos = [], [], []
    boxes = gt_boxes.float()
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    centers = (boxes[:, :2] + boxes[:, 2:]) / 2                                     # [N,2]
    radius2 = torch.clamp(delta ** 2 * 2 * area, min=min_radius ** 2)               # [N]
    # positive locations: box-major, level-minor (centernet.py:463-473)
    crit_box = ((boxes[:, 2:] - boxes[:, :2]) ** 2).sum(dim=1) ** 0.5 / 2
    base = 0
    bases = []
    for (h, w) in level_hw:
        bases.append(base)
        base += h * w
    for n in range(N):
        for l, s in enumerate(strides):
            lo, hi = sizes_of_interest[l]
            if float(crit_box[n]) >= lo and float(crit_box[n]) <= hi:
                ci = (centers[n] / float(s)).long()
                pos.append(bases[l] + int(ci[1]) * level_hw[l][1] + int(ci[0]))
    for l, ((h, w), s) in enumerate(zip(level_hw, strides)):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) * s + s // 2, torch.arange(w, dtype=torch.float32) * s + s // 2,
                                indexing="ij")
        gx, gy = xs.reshape(-1, 1), ys.reshape(-1, 1)                               # [m,1]
        m = gx.shape[0]
        if N == 0:
            regs.append(torch.full((m, 4), -float(CENTERNET_INF)) / float(s))
            heats.append(torch.zeros((m,)))
            continue
        lt = gx - boxes[:, 0].view(1, N)
        tt = gy - boxes[:, 1].view(1, N)
        rt = boxes[:, 2].view(1, N) - gx
        bt = boxes[:, 3].view(1, N) - gy
        ltrb = torch.stack([lt, tt, rt, bt], dim=2)                                 # [m,N,4]
        cdx = (centers[:, 0].view(1, N) / s).int().float() * s + s / 2
        cdy = (centers[:, 1].view(1, N) / s).int().float() * s + s / 2
        is_peak = ((gx - cdx) ** 2 + (gy - cdy) ** 2) == 0
        in_box = ltrb.min(dim=2)[0] > 0
        c3 = ((gx - cdx).abs() <= s) & ((gy - cdy).abs() <= s) & in_box
        crit = ((lt + rt) ** 2 + (tt + bt) ** 2) ** 0.5 / 2
        lo, hi = sizes_of_interest[l]
        mask = c3 & (crit >= lo) & (crit <= hi)
        dist2 = (gx - centers[:, 0].view(1, N)) ** 2 + (gy - centers[:, 1].view(1, N)) ** 2
        dist2[is_peak] = 0
        wd = dist2 / radius2.view(1, N)
        d = wd.clone()
        d[~mask] = CENTERNET_INF * 1.0
        md, mi = d.min(dim=1)
        r = ltrb[torch.arange(m), mi].clone()
        r[md == CENTERNET_INF] = -float(CENTERNET_INF)
        regs.append(r / float(s))
        hm = torch.exp(-wd.min(dim=1)[0])
        hm[hm < 1e-4] = 0
        heats.append(hm)
    return torch.tensor(pos, dtype=torch.int64), torch.cat(regs), torch.cat(heats)


# ----------------------------------------------------------------------------------------------------------------------------------
# The ROI heads' half of the training forward: `DeticCascadeROIHeads.forward` / `_forward_box` in training mode with ann_type 'box'
# (`Detic/detic/modeling/roi_heads/detic_roi_heads.py:226-247, 88-147, 306-326`).  The detectron2 pieces they call are not in the
# reference tree and are restated from their published definitions ("unpinned", as SURVEY Appendix A says of detectron2):
# `pairwise_iou` (structures/boxes.py), `Matcher` (modeling/matcher.py), `subsample_labels` (modeling/sampling.py),
# `add_ground_truth_to_proposals` (modeling/proposal_generator/proposal_utils.py), `ROIHeads.label_and_sample_proposals` /
# `_sample_proposals` (modeling/roi_heads/roi_heads.py), `CascadeROIHeads._match_and_label_boxes` (modeling/roi_heads/cascade_rcnn.py).
# The random draw of `subsample_labels` (torch.randperm) is restated as a selection by given random keys: the uniform random subset
# of the same size (the reference's RNG stream is not reproducible across devices).
# ----------------------------------------------------------------------------------------------------------------------------------
def pairwise_iou(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    """detectron2 `pairwise_iou`: [N,4] x [M,4] -> [N,M]."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    wh = torch.min(boxes1[:, None, 2:], boxes2[:, 2:]) - torch.max(boxes1[:, None, :2], boxes2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter), torch.zeros(1, dtype=inter.dtype))


def match_label(boxes: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor, iou_thresh: float, num_classes: int):
    """Matcher([iou_thresh], [0, 1], allow_low_quality_matches=False) on pairwise_iou(gt, proposals) + the labelling of
    `_sample_proposals` / `_match_and_label_boxes` -> (matched_idx [R], matched_iou [R], classes [R] (background = num_classes),
    matched gt boxes [R,4]; zeros without ground truth)."""
    R = boxes.shape[0]
    if gt_boxes.shape[0] == 0:
        return (torch.zeros(R, dtype=torch.int64), torch.zeros(R), torch.full((R,), num_classes, dtype=torch.int64), torch.zeros(R, 4))
    q = pairwise_iou(gt_boxes, boxes)
    vals, idx = q.max(dim=0)
    labels = torch.ones(R, dtype=torch.int8)
    for lab, low, high in ((0, -float("inf"), iou_thresh), (1, iou_thresh, float("inf"))):
        labels[(vals >= low) & (vals < high)] = lab
    cls = gt_classes[idx].clone()
    cls[labels == 0] = num_classes
    return idx, vals, cls, gt_boxes[idx]


def sample_by_keys(classes: torch.Tensor, keys: torch.Tensor, num_classes: int, batch: int, positive_fraction: float) -> torch.Tensor:
    """`subsample_labels(classes, batch, positive_fraction, bg_label=num_classes)` with the random subset chosen by smallest key;
    -> sampled row indices, foreground first, each kind in ascending row order."""
    pos = torch.nonzero((classes != -1) & (classes != num_classes)).squeeze(1)
    neg = torch.nonzero(classes == num_classes).squeeze(1)
    num_pos = min(pos.numel(), int(batch * positive_fraction))
    num_neg = min(neg.numel(), batch - num_pos)

    def pick(rows, n):
        order = sorted(rows.tolist(), key=lambda r: (float(keys[r]), r))[:n]
        return torch.tensor(sorted(order), dtype=torch.int64)
    return torch.cat([pick(pos, num_pos), pick(neg, num_neg)])


class _ScaleGradient(torch.autograd.Function):
    """detectron2 `_ScaleGradient` (modeling/roi_heads/cascade_rcnn.py): identity forward, gradient x scale."""
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


GT_PROPOSAL_LOGIT = 23.025850847100816        # add_ground_truth_to_proposals: log((1 - 1e-10) / (1 - (1 - 1e-10)))


def cascade_training_losses(feats, prop_boxes: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor, sd, cfg, image_hw,
                            keys: torch.Tensor, ious=(0.6, 0.7, 0.8), batch: int = 512, positive_fraction: float = 0.25,
                            smooth_l1_beta: float = 0.0, append_gt: bool = True):
    """`DeticCascadeROIHeads.forward` in training, ann_type 'box', no gt_masks (the MP3D loader provides none:
    `_get_empty_mask_loss`, detic_roi_heads.py:246-249,297-303) -> ({loss name: scalar}, per-stage intermediates).
    feats: P3..P5 NCHW; prop_boxes [R,4]; keys [R + G] uniform random (one per proposal after the ground truth is appended)."""
    from . import model as M, ops as O
    boxes = torch.cat([prop_boxes, gt_boxes]) if append_gt else prop_boxes          # label_and_sample_proposals (:232)
    _, _, cls, gtb = match_label(boxes, gt_boxes, gt_classes, ious[0], cfg.num_classes)
    rows = sample_by_keys(cls, keys, cfg.num_classes, batch, positive_fraction)
    sampled_rows = rows
    boxes, cls, gtb = boxes[rows], cls[rows], gtb[rows]
    losses, stages = {}, []
    for k in range(3):
        if k > 0:                                                                   # _create_proposals_from_boxes (:306-326)
            boxes = O.clip_boxes(boxes, image_hw)
            keep = (boxes[:, 2] - boxes[:, 0] > 0) & (boxes[:, 3] - boxes[:, 1] > 0)    # Boxes.nonempty
            boxes = boxes[keep]
            _, _, cls, gtb = match_label(boxes, gt_boxes, gt_classes, ious[k], cfg.num_classes)     # _match_and_label_boxes (:115)
        pooled = _ScaleGradient.apply(O.roi_pool(list(feats[:3]), boxes, 7), 1.0 / 3)   # _run_stage (:328-349)
        logits, deltas, _ = M.box_head_stage(pooled, sd, k, cfg)
        with torch.no_grad():                                                       # the stage's ReLU patterns (tests count knife-edge flips)
            a1 = F.relu(F.linear(pooled.flatten(1), sd[f"roi_heads.box_head.{k}.fc1.weight"], sd[f"roi_heads.box_head.{k}.fc1.bias"]))
            a2 = F.relu(F.linear(a1, sd[f"roi_heads.box_head.{k}.fc2.weight"], sd[f"roi_heads.box_head.{k}.fc2.bias"]))
            ab = F.relu(F.linear(a2, sd[f"roi_heads.box_predictor.{k}.bbox_pred.0.weight"], sd[f"roi_heads.box_predictor.{k}.bbox_pred.0.bias"]))
        losses[f"loss_cls_stage{k}"] = sigmoid_cross_entropy_loss(logits, cls)
        losses[f"loss_box_reg_stage{k}"] = box_reg_loss(boxes, gtb, deltas, cls, cfg.num_classes, M.CASCADE_WEIGHTS[k], smooth_l1_beta)
        stages.append(dict(boxes=boxes, classes=cls, gt_boxes=gtb, logits=logits, deltas=deltas, h1=a1, h2=a2, hb=ab, rows=sampled_rows))
        boxes = O.apply_deltas(deltas.detach(), boxes, M.CASCADE_WEIGHTS[k])       # predict_boxes (:124)
    losses["loss_mask"] = torch.zeros(())                                           # _get_empty_mask_loss, MASK_ON
    return losses, stages
