#!/usr/bin/env python
"""Golden fixture of the ROI heads' TRAINING forward: runs the reference's own `DeticCascadeROIHeads.forward` in training mode
(`Detic/detic/modeling/roi_heads/detic_roi_heads.py:226-249`) -- its `_forward_box` training branch (:88-147: stage chaining,
`_stage{k}` loss names), `_create_proposals_from_boxes` (:306-326, with the training-only non-empty filter), `_run_stage` (:328-349),
`_get_empty_mask_loss` (:297-303) -- with three real `DeticFastRCNNOutputLayers` (forward :437-466 and `losses` :157-197 with
`sigmoid_cross_entropy_loss` / `box_reg_loss`) each holding a real `ZeroShotClassifier`.

detectron2 is not installed.  INJECTED from their restatements (`oracle/losses.py`, `oracle/ops.py`), exactly as `gen_golden.py` /
`gen_golden_losses.py` do for the inference fixtures: `label_and_sample_proposals` and `_match_and_label_boxes` (pairwise_iou, Matcher,
subsample_labels by given random keys, add_ground_truth_to_proposals), `Boxes.nonempty`, ROIPooler, the FC box head,
`predict_boxes`, `Box2BoxTransform.get_deltas`, `smooth_l1_loss`, `nonzero_tuple`, `cat`, `get_event_storage`.  What the fixture pins
is the reference's OWN control flow and arithmetic around them.

    python tests/golden/gen_golden_cascade_training.py        # needs /root/reference; writes tests/golden/cascade_training.npz
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

IOUS = (0.6, 0.7, 0.8)                 # ROI_BOX_CASCADE_HEAD.IOUS of the recurrent yaml
BATCH, FRACTION = 32, 0.25            # a batch smaller than the proposal list, so that the sampling decides


def training_case():
    """Ground truth for `_inputs.cascade_case()`'s 48 proposals: eight of the proposals' own boxes, slightly moved (so that matches of
    every quality exist), classes, and the sampling keys."""
    import _inputs as I
    feats, boxes, scores = I.cascade_case()
    g = torch.Generator().manual_seed(77)
    pick = torch.randperm(boxes.shape[0], generator=g)[:8]
    gt = boxes[pick] + torch.randn((8, 4), generator=g) * 2.0
    H, W = I.CASCADE_HW
    gt = torch.stack([gt[:, 0].clamp(0, W - 8), gt[:, 1].clamp(0, H - 8), gt[:, 2].clamp(8, W), gt[:, 3].clamp(8, H)], dim=1)
    gt[:, 2:] = torch.maximum(gt[:, 2:], gt[:, :2] + 4.0)
    gc = torch.randint(0, 20, (8,), generator=g)
    keys = torch.rand((boxes.shape[0] + 8,), generator=g)
    return feats, boxes, scores, gt, gc, keys


def main():
    import gen_golden as G
    import _inputs as I
    from oracle import losses as OL
    from oracle import ops as O
    G.install_shim()
    mods = G.load_reference_modules()
    RH, FR, Z = mods["roi_heads"], mods["fast_rcnn"], mods["zs"].ZeroShotClassifier
    FR.nonzero_tuple = lambda x: torch.nonzero(x, as_tuple=True)
    FR.smooth_l1_loss = lambda a, b, beta, reduction="sum": OL.smooth_l1_sum(a, b, beta)
    FR.cat = lambda ts, dim=0: torch.cat(list(ts), dim=dim)
    G._Boxes.nonempty = lambda self, threshold=0.0: ((self.tensor[:, 2] - self.tensor[:, 0]) > threshold) & \
        ((self.tensor[:, 3] - self.tensor[:, 1]) > threshold)                   # detectron2 Boxes.nonempty

    class _Storage:
        def name_scope(self, name):
            return contextlib.nullcontext()
    RH.get_event_storage = lambda: _Storage()

    feats, boxes, scores, gt, gc, keys = training_case()
    sd = I.cascade_weights()
    clip_path = os.path.join(G.DETIC, "datasets", "metadata", "mp3d_clip.npy")
    weights = ((10.0, 10.0, 5.0, 5.0), (20.0, 20.0, 10.0, 10.0), (30.0, 30.0, 15.0, 15.0))
    heads, preds = [], []
    for k in range(3):
        head = nn.Sequential(nn.Flatten(), nn.Linear(12544, 1024), nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU())
        shape = types.SimpleNamespace(channels=1024, width=None, height=None)
        pred = FR.DeticFastRCNNOutputLayers(shape, box2box_weights=weights[k], num_classes=20, test_score_thresh=0.02,
                                            test_nms_thresh=0.5, test_topk_per_image=300, mult_proposal_score=True,
                                            cls_score=Z(shape, num_classes=20, zs_weight_path=clip_path),
                                            use_sigmoid_ce=True, use_zeroshot_cls=True)
        # what detectron2's FastRCNNOutputLayers.__init__ leaves for the losses (SMOOTH_L1_BETA 0, smooth_l1, the stage's transform)
        pred.box_reg_loss_type, pred.smooth_l1_beta = "smooth_l1", 0.0
        pred.box2box_transform = types.SimpleNamespace(get_deltas=lambda a, b, w=weights[k]: OL.get_deltas(a, b, w))
        with torch.no_grad():
            head[1].weight.copy_(sd[f"roi_heads.box_head.{k}.fc1.weight"]); head[1].bias.copy_(sd[f"roi_heads.box_head.{k}.fc1.bias"])
            head[3].weight.copy_(sd[f"roi_heads.box_head.{k}.fc2.weight"]); head[3].bias.copy_(sd[f"roi_heads.box_head.{k}.fc2.bias"])
            p = f"roi_heads.box_predictor.{k}"
            pred.cls_score.linear.weight.copy_(sd[f"{p}.cls_score.linear.weight"]); pred.cls_score.linear.bias.copy_(sd[f"{p}.cls_score.linear.bias"])
            pred.bbox_pred[0].weight.copy_(sd[f"{p}.bbox_pred.0.weight"]); pred.bbox_pred[0].bias.copy_(sd[f"{p}.bbox_pred.0.bias"])
            pred.bbox_pred[2].weight.copy_(sd[f"{p}.bbox_pred.2.weight"]); pred.bbox_pred[2].bias.copy_(sd[f"{p}.bbox_pred.2.bias"])
        heads.append(head)
        preds.append(pred)

    stage_in = []

    def pooler(features, box_lists):
        stage_in.append(box_lists[0].tensor.clone())
        return O.roi_pool(features, box_lists[0].tensor, 7)

    sampled = {}

    def label_and_sample_proposals(proposals, targets):
        """detectron2 ROIHeads.label_and_sample_proposals restated on the oracle's functions (PROPOSAL_APPEND_GT)."""
        out = []
        for p, t in zip(proposals, targets):
            b = torch.cat([p.proposal_boxes.tensor, t.gt_boxes.tensor])
            logits = torch.cat([p.objectness_logits, torch.full((len(t),), OL.GT_PROPOSAL_LOGIT)])
            _, _, cls, gtb = OL.match_label(b, t.gt_boxes.tensor, t.gt_classes, IOUS[0], 20)
            rows = OL.sample_by_keys(cls, keys, 20, BATCH, FRACTION)
            sampled["rows"] = rows
            q = G._Instances(p.image_size, proposal_boxes=G._Boxes(b[rows]), objectness_logits=logits[rows], gt_classes=cls[rows])
            if len(t) > 0:
                q.gt_boxes = G._Boxes(gtb[rows])
            out.append(q)
        return out

    def match_and_label_boxes(proposals, stage, targets):
        """detectron2 CascadeROIHeads._match_and_label_boxes restated."""
        for p, t in zip(proposals, targets):
            _, _, cls, gtb = OL.match_label(p.proposal_boxes.tensor, t.gt_boxes.tensor, t.gt_classes, IOUS[stage], 20)
            p.gt_classes = cls
            p.gt_boxes = G._Boxes(gtb)
        return proposals

    rh = G._bare(RH.DeticCascadeROIHeads, mult_proposal_score=True, add_feature_to_prop=True, one_class_per_proposal=False,
                 num_cascade_stages=3, box_in_features=["p3", "p4", "p5"], box_pooler=pooler, box_head=nn.ModuleList(heads),
                 box_predictor=nn.ModuleList(preds), mask_on=True, mask_weight=1.0, with_image_labels=False,
                 label_and_sample_proposals=label_and_sample_proposals, _match_and_label_boxes=match_and_label_boxes)
    rh.train()
    prop = G._Instances(I.CASCADE_HW, proposal_boxes=G._Boxes(boxes), scores=scores, objectness_logits=scores)
    target = G._Instances(I.CASCADE_HW, gt_boxes=G._Boxes(gt), gt_classes=gc)
    with torch.no_grad():
        proposals, losses = rh.forward(None, {"p3": feats[0], "p4": feats[1], "p5": feats[2]}, [prop], [target])
    assert sorted(losses) == sorted([f"loss_{n}_stage{k}" for k in range(3) for n in ("cls", "box_reg")] + ["loss_mask"]), sorted(losses)
    assert len(stage_in) == 3 and stage_in[0].shape[0] == BATCH
    out = {k: np.float64(float(v)) for k, v in losses.items()}
    out.update(gt_boxes=gt.numpy(), gt_classes=gc.numpy().astype(np.int64), keys=keys.numpy(), sampled_rows=sampled["rows"].numpy(),
               stage0_boxes=stage_in[0].numpy(), stage1_boxes=stage_in[1].numpy(), stage2_boxes=stage_in[2].numpy(),
               sampled_classes=proposals[0].gt_classes.numpy().astype(np.int64), batch=np.int64(BATCH), ious=np.array(IOUS))
    path = os.path.join(HERE, "cascade_training.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: round(float(v), 6) for k, v in losses.items()}, [tuple(s.shape) for s in stage_in],
          "foreground", int((proposals[0].gt_classes < 20).sum()))


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
