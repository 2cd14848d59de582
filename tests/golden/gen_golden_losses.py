#!/usr/bin/env python
"""Golden vectors of the proposal generator's training losses  --  runs ONLY in the development container (needs /root/reference).

Loads the reference's OWN loss modules from where they lie (`Detic/third_party/CenterNet2/centernet/modeling/layers/
heatmap_focal_loss.py`, `iou_loss.py`: plain torch, no detectron2) and evaluates `binary_heatmap_focal_loss_jit` and
`IOULoss('giou')` exactly as `CenterNet.losses` calls them (`.../dense_heads/centernet.py:283-313`) on seeded inputs of a five-level
pyramid, with autograd's gradients with respect to the agnostic logits and the regression predictions.  Stores inputs, the three
losses and the gradients in `tests/golden/centernet_loss.npz`; `main_box` does the same for the cascade's box-head losses
(`tests/golden/fast_rcnn_loss.npz`).  No reference source or bytecode is copied.

    python tests/golden/gen_golden_losses.py
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
LAYERS = "/root/reference/Detic/third_party/CenterNet2/centernet/modeling/layers"


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(LAYERS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_inputs(seed: int = 0):
    """A five-level pyramid of a 128x160 image: M = 427 positions; 23 positive locations (one of them listed twice: two objects with
    one centre), Gaussian-like heatmap peaks around them, regression targets on 60 positions (-INF elsewhere, centernet.py:283)."""
    g = torch.Generator().manual_seed(seed)
    shapes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    M = sum(h * w for h, w in shapes)
    logits = torch.randn((M,), generator=g) * 2.5
    logits[::37] = 9.5                        # sigmoid above 1 - clamp
    logits[5::41] = -9.5                      # and below the clamp
    heat = torch.rand((M,), generator=g) ** 6
    pos = torch.randperm(M, generator=g)[:22].sort().values
    pos = torch.cat([pos, pos[3:4]])
    heat[pos] = 1.0
    reg_pred = torch.rand((M, 4), generator=g) * 40
    reg_pred[torch.rand((M, 4), generator=g) < 0.1] = 0.0     # ReLU zeros
    reg_targets = torch.full((M, 4), -1e8)                     # INF of the target assignment (centernet.py: INF = 100000000)
    rows = torch.randperm(M, generator=g)[:60]
    reg_targets[rows] = torch.rand((60, 4), generator=g) * 50 + 0.5
    return shapes, logits, heat, pos, reg_pred, reg_targets


def main():
    hfl = _load("heatmap_focal_loss")
    iou = _load("iou_loss")
    shapes, logits, heat, pos, reg_pred, reg_targets = make_inputs(0)
    cfg = dict(alpha=0.25, beta=4.0, gamma=2.0, sigmoid_clamp=1e-4, ignore_high_fp=0.85, pos_weight=0.5, neg_weight=0.5, reg_weight=1.0)
    z = logits.clone().requires_grad_()
    r = reg_pred.clone().requires_grad_()
    # as CenterNet.losses: reg_inds, weight map (not_norm_reg: ones), norms on one rank
    num_pos_avg = max(float(pos.numel()), 1.0)
    reg_inds = torch.nonzero(reg_targets.max(dim=1)[0] >= 0).squeeze(1)
    w = heat[reg_inds] * 0 + 1
    reg_norm = max(float(w.sum()), 1)
    loc = cfg["reg_weight"] * iou.IOULoss("giou")(r[reg_inds], reg_targets[reg_inds], w, reduction="sum") / reg_norm
    # the reference applies sigmoid_ in place on `agn_hm_pred.float()`: hand it a differentiable copy (z * 1) so autograd still works
    ap, an = hfl.binary_heatmap_focal_loss_jit((z * 1.0).float(), heat.float(), pos, alpha=cfg["alpha"], beta=cfg["beta"],
                                               gamma=cfg["gamma"], sigmoid_clamp=cfg["sigmoid_clamp"],
                                               ignore_high_fp=cfg["ignore_high_fp"])
    ap = cfg["pos_weight"] * ap / num_pos_avg
    an = cfg["neg_weight"] * an / num_pos_avg
    (loc + ap + an).backward()
    out = dict(shapes=np.array(shapes, np.int32), logits=logits.numpy(), heat=heat.numpy(), pos_inds=pos.numpy().astype(np.int64),
               reg_pred=reg_pred.numpy(), reg_targets=reg_targets.numpy(), loss_loc=np.float64(loc.item()),
               loss_agn_pos=np.float64(ap.item()), loss_agn_neg=np.float64(an.item()), grad_logits=z.grad.numpy(),
               grad_reg=r.grad.numpy(), num_pos_avg=np.float64(num_pos_avg), reg_norm=np.float64(reg_norm),
               **{"cfg_" + k: np.float64(v) for k, v in cfg.items()})
    path = os.path.join(HERE, "centernet_loss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: float(out[k]) for k in ("loss_loc", "loss_agn_pos", "loss_agn_neg")})


def make_box_inputs(seed: int = 1, B: int = 96, C: int = 20):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn((B, C + 1), generator=g) * 3
    gt = torch.randint(0, C + 1, (B,), generator=g)
    gt[::3] = C                                                   # background rows
    xy = torch.rand((B, 2), generator=g) * 300
    wh = torch.rand((B, 2), generator=g) * 200 + 4
    prop = torch.cat([xy, xy + wh], dim=1)
    jit = (torch.rand((B, 4), generator=g) - 0.5) * 30
    gtb = prop + jit
    gtb[:, 2:] = torch.maximum(gtb[:, 2:], gtb[:, :2] + 2)
    deltas = torch.randn((B, 4), generator=g) * 0.5
    cw = (torch.rand((C,), generator=g) < 0.7).float()            # a federated-loss mask
    return logits, gt, prop, gtb, deltas, cw


def main_box():
    """`DeticFastRCNNOutputLayers.sigmoid_cross_entropy_loss` / `box_reg_loss` (detic_fast_rcnn.py:200-233, 270-303) run as unbound
    methods of the reference class on a stand-in `self` (the attributes `__init__` would set for USE_SIGMOID_CE, class-agnostic
    regression, smooth_l1 beta 0, one cascade stage's weights).  detectron2 / fvcore are absent: `nonzero_tuple`, `smooth_l1_loss` and
    `Box2BoxTransform.get_deltas` are injected from their restatements in `oracle/losses.py`; what the fixture pins is the
    reference's own part -- target construction, the class-weight mask, foreground selection, the normalisations."""
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    sys.path.insert(0, HERE)
    import gen_golden as G
    from oracle import losses as OL
    G.install_shim()
    mods = G.load_reference_modules()
    fr = mods["fast_rcnn"]
    fr.nonzero_tuple = lambda x: torch.nonzero(x, as_tuple=True)
    fr.smooth_l1_loss = lambda a, b, beta, reduction="sum": OL.smooth_l1_sum(a, b, beta)
    weights = (20.0, 20.0, 10.0, 10.0)                             # second cascade stage (ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS)
    C = 20
    logits, gt, prop, gtb, deltas, cw = make_box_inputs()
    out = {}
    for tag, fw in (("plain", None), ("fed", cw)):
        self_ = types.SimpleNamespace(num_classes=C, use_fed_loss=False, ignore_zero_cats=fw is not None,
                                      freq_weight=None if fw is None else fw * 1.0, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0,
                                      box2box_transform=types.SimpleNamespace(get_deltas=lambda a, b: OL.get_deltas(a, b, weights)))
        z = logits.clone().requires_grad_()
        d = deltas.clone().requires_grad_()
        lc = fr.DeticFastRCNNOutputLayers.sigmoid_cross_entropy_loss(self_, z, gt)
        lb = fr.DeticFastRCNNOutputLayers.box_reg_loss(self_, prop, gtb, d, gt, num_classes=C)
        (lc + lb).backward()
        out.update({f"{tag}_loss_cls": np.float64(lc.item()), f"{tag}_loss_box_reg": np.float64(lb.item()),
                    f"{tag}_grad_logits": z.grad.numpy(), f"{tag}_grad_deltas": d.grad.numpy()})
    out.update(logits=logits.numpy(), gt_classes=gt.numpy().astype(np.int64), proposal_boxes=prop.numpy(), gt_boxes=gtb.numpy(),
               deltas=deltas.numpy(), class_weight=cw.numpy(), box_weights=np.array(weights, np.float64), num_classes=np.int64(C))
    path = os.path.join(HERE, "fast_rcnn_loss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: float(v) for k, v in out.items() if "loss" in k})


class _GtBoxes:
    def __init__(self, t):
        self.tensor = t

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def make_gt_boxes(seed: int = 2):
    """Ground-truth boxes of a 256x320 image: small / medium / large (all five size ranges of MODEL.CENTERNET.SOI are hit), two boxes
    with one centre, a centre exactly on a grid point, a box reaching outside the image."""
    g = torch.Generator().manual_seed(seed)
    boxes = [[20.0, 30.0, 52.0, 70.0], [100.5, 40.25, 180.0, 150.0], [4.0, 4.0, 316.0, 252.0], [60.0, 60.0, 68.0, 66.0],
             [200.0, 100.0, 310.0, 240.0], [210.0, 110.0, 300.0, 230.0], [92.0, 92.0, 100.0, 100.0], [-10.0, 180.0, 90.0, 270.0],
             [120.0, 8.0, 250.0, 120.0], [30.0, 140.0, 190.0, 250.0]]
    extra = torch.rand((6, 2), generator=g) * torch.tensor([250.0, 200.0])
    wh = torch.rand((6, 2), generator=g) * 90 + 6
    boxes = torch.cat([torch.tensor(boxes), torch.cat([extra, extra + wh], dim=1)])
    classes = torch.randint(0, 20, (boxes.shape[0],), generator=g)
    return boxes, classes


def main_targets():
    """`CenterNet._get_ground_truth` + `_get_label_inds` (centernet.py:342-479) and every helper they call (`compute_grids`,
    `assign_fpn_level`, `assign_reg_fpn`, `_get_reg_targets`, `_create_agn_heatmaps_from_dist`, `get_center3x3`, `_transpose`) run as
    the reference's own code on a stand-in `self` carrying the configuration's attributes (ONLY_PROPOSAL, default SOI / strides /
    HM_MIN_OVERLAP / MIN_RADIUS); detectron2's `cat` is torch.cat.  One image with 16 boxes and one image without objects."""
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    sys.path.insert(0, HERE)
    import gen_golden as G
    if "centernet.modeling.dense_heads.centernet" not in sys.modules:
        G.install_shim()
        G.load_reference_modules()
    cn = sys.modules["centernet.modeling.dense_heads.centernet"]
    cn.cat = torch.cat
    CN = cn.CenterNet
    self_ = types.SimpleNamespace(strides=[8, 16, 32, 64, 128], num_classes=20, only_proposal=True, more_pos=False, not_clamp_box=False,
                                  sizes_of_interest=[[0, 80], [64, 160], [128, 320], [256, 640], [512, 10000000]], min_radius=4,
                                  delta=(1 - 0.8) / (1 + 0.8))
    for name in ("compute_grids", "_get_label_inds", "assign_fpn_level", "assign_reg_fpn", "_get_reg_targets",
                 "_create_agn_heatmaps_from_dist", "get_center3x3", "_get_ground_truth"):
        setattr(self_, name, types.MethodType(getattr(CN, name), self_))
    H, W = 256, 320
    shapes = [(H // s + (1 if H % s else 0), W // s + (1 if W % s else 0)) for s in self_.strides]
    feats = [torch.zeros((1, 1, h, w)) for h, w in shapes]
    grids = self_.compute_grids(feats)
    shapes_per_level = grids[0].new_tensor(shapes)
    boxes, classes = make_gt_boxes()
    out = dict(image_hw=np.array([H, W], np.int32), shapes=np.array(shapes, np.int32), gt_boxes=boxes.numpy(),
               gt_classes=classes.numpy().astype(np.int64), grids=torch.cat(grids).numpy())
    for tag, b, c in (("full", boxes, classes), ("empty", boxes[:0], classes[:0])):
        inst = [types.SimpleNamespace(gt_boxes=_GtBoxes(b.clone()), gt_classes=c.clone())]
        pos_inds, labels, reg_targets, hms = self_._get_ground_truth([x.clone() for x in grids], shapes_per_level, inst)
        out.update({f"{tag}_pos_inds": pos_inds.numpy().astype(np.int64), f"{tag}_labels": labels.numpy().astype(np.int64),
                    f"{tag}_reg_targets": reg_targets.numpy(), f"{tag}_heatmap": hms.numpy()})
    path = os.path.join(HERE, "centernet_targets.npz")
    np.savez_compressed(path, **out)
    rt = out["full_reg_targets"]
    print("wrote", path, "positions", rt.shape[0], "positives", len(out["full_pos_inds"]), "regression rows",
          int((rt.max(axis=1) >= 0).sum()), "heatmap > 0", int((out["full_heatmap"] > 0).sum()))


if __name__ == "__main__":
    main()
    main_box()
    main_targets()
