"""The oracle's restatement of the proposal generator's training losses against the reference's own functions
(`tests/golden/centernet_loss.npz`, made by `tests/golden/gen_golden_losses.py` from heatmap_focal_loss.py / iou_loss.py as
`CenterNet.losses` calls them, centernet.py:241-318).  CPU."""
import os

import numpy as np
import torch

from oracle import losses as OL

GOLD = os.path.join(os.path.dirname(__file__), "golden", "centernet_loss.npz")


def load_fixture():
    z = np.load(GOLD)
    t = lambda k: torch.from_numpy(z[k])
    cfg = {k[4:]: float(z[k]) for k in z.files if k.startswith("cfg_")}
    return z, t, cfg


def test_oracle_losses_match_the_reference_functions():
    z, t, cfg = load_fixture()
    logits = t("logits").clone().requires_grad_()
    reg = t("reg_pred").clone().requires_grad_()
    out = OL.centernet_proposal_losses(logits, reg, t("heat"), t("reg_targets"), t("pos_inds"), **cfg)
    for k, name in (("loss_loc", "loss_centernet_loc"), ("loss_agn_pos", "loss_centernet_agn_pos"), ("loss_agn_neg", "loss_centernet_agn_neg")):
        assert abs(out[name].item() - float(z[k])) <= 1e-6 * abs(float(z[k])), k
    sum(out.values()).backward()
    assert float((logits.grad - t("grad_logits")).abs().max()) <= 1e-7
    assert float((reg.grad - t("grad_reg")).abs().max()) <= 1e-7
    # the fixture exercises what it claims to: clamped logits (zero gradient), a duplicated positive, ReLU zeros, ignored positions
    assert int((t("grad_logits") == 0).sum()) > 10 and len(set(z["pos_inds"].tolist())) < len(z["pos_inds"])
    assert int((t("reg_targets").max(dim=1)[0] >= 0).sum()) == 60


GOLD_BOX = os.path.join(os.path.dirname(__file__), "golden", "fast_rcnn_loss.npz")


def load_box_fixture():
    z = np.load(GOLD_BOX)
    return z, (lambda k: torch.from_numpy(z[k]))


def test_oracle_box_head_losses_match_the_reference_methods():
    """`oracle.losses.sigmoid_cross_entropy_loss` / `box_reg_loss` against the reference's own methods
    (detic_fast_rcnn.py:200-233, 270-303; gen_golden_losses.py::main_box), with and without a class-weight mask."""
    z, t = load_box_fixture()
    C = int(z["num_classes"])
    weights = tuple(float(v) for v in z["box_weights"])
    for tag, cw in (("plain", None), ("fed", t("class_weight"))):
        logits = t("logits").clone().requires_grad_()
        deltas = t("deltas").clone().requires_grad_()
        lc = OL.sigmoid_cross_entropy_loss(logits, t("gt_classes"), cw)
        lb = OL.box_reg_loss(t("proposal_boxes"), t("gt_boxes"), deltas, t("gt_classes"), C, weights, 0.0)
        assert abs(lc.item() - float(z[f"{tag}_loss_cls"])) <= 1e-6 * float(z[f"{tag}_loss_cls"])
        assert abs(lb.item() - float(z[f"{tag}_loss_box_reg"])) <= 1e-6 * float(z[f"{tag}_loss_box_reg"])
        (lc + lb).backward()
        assert float((logits.grad - t(f"{tag}_grad_logits")).abs().max()) <= 1e-8
        assert float((deltas.grad - t(f"{tag}_grad_deltas")).abs().max()) <= 1e-8
    assert int((t("gt_classes") == C).sum()) >= 32 and float(t("fed_grad_logits")[:, C].abs().max()) == 0.0


GOLD_TGT = os.path.join(os.path.dirname(__file__), "golden", "centernet_targets.npz")


def test_oracle_target_assignment_matches_the_reference():
    """`oracle.losses.centernet_targets` against the reference's own `_get_ground_truth` / `_get_label_inds` (centernet.py:342-479;
    gen_golden_losses.py::main_targets): positive locations, regression targets and the agnostic heatmap, bit for bit; and the
    image without objects."""
    z = np.load(GOLD_TGT)
    shapes = [tuple(x) for x in z["shapes"].tolist()]
    for tag, boxes in (("full", torch.from_numpy(z["gt_boxes"])), ("empty", torch.zeros((0, 4)))):
        pos, reg, heat = OL.centernet_targets(boxes, shapes)
        assert pos.tolist() == z[f"{tag}_pos_inds"].tolist(), tag
        assert np.array_equal(reg.numpy(), z[f"{tag}_reg_targets"]), tag
        assert np.array_equal(heat.numpy(), z[f"{tag}_heatmap"][:, 0]), tag
    assert len(z["full_pos_inds"]) > len(z["gt_boxes"]) and int((z["full_reg_targets"].max(axis=1) >= 0).sum()) > 100


def test_oracle_matcher_and_sampler_hand_cases():
    """The restated detectron2 pieces of the ROI heads' training forward (`oracle/losses.py`: pairwise_iou, Matcher + labelling,
    subsample_labels by keys) on cases worked by hand; unpinned (detectron2 is not in the reference tree)."""
    import pytest
    gt = torch.tensor([[0.0, 0.0, 10.0, 10.0], [20.0, 20.0, 40.0, 40.0], [0.0, 0.0, 10.0, 10.0]])
    gc = torch.tensor([3, 7, 5])
    props = torch.tensor([[0.0, 0.0, 10.0, 10.0],        # IoU 1 with objects 0 and 2: the first wins
                          [0.0, 0.0, 10.0, 5.0],         # IoU 0.5 with object 0
                          [20.0, 20.0, 40.0, 36.0],      # IoU 0.8 with object 1
                          [50.0, 50.0, 60.0, 60.0],      # no overlap
                          [10.0, 10.0, 20.0, 20.0]])     # touches objects 0 and 1: intersection 0
    iou = OL.pairwise_iou(gt, props)
    assert torch.allclose(iou[:, 1], torch.tensor([0.5, 0.0, 0.5])) and float(iou[1, 2]) == pytest.approx(0.8) and float(iou[:, 4].max()) == 0.0
    idx, vals, cls, gtb = OL.match_label(props, gt, gc, 0.6, 20)
    assert idx.tolist()[:3] == [0, 0, 1] and cls.tolist() == [3, 20, 7, 20, 20]
    assert torch.equal(gtb[2], gt[1]) and vals.tolist()[3:] == [0.0, 0.0]
    assert OL.match_label(props, gt, gc, 0.5, 20)[2].tolist() == [3, 3, 7, 20, 20]            # >= at the threshold is foreground
    e = OL.match_label(props, torch.zeros((0, 4)), torch.zeros((0,), dtype=torch.int64), 0.6, 20)
    assert e[2].tolist() == [20] * 5 and float(e[3].abs().max()) == 0.0
    # sampling: 2 of the 3 foreground rows (batch 8 x 1/4), then 6 background rows by smallest key; -1 rows never
    cls = torch.tensor([20, 1, 20, -1, 4, 20, 20, 20, 9, 20, 20, 20])
    keys = torch.tensor([0.9, 0.3, 0.1, 0.0, 0.2, 0.5, 0.5, 0.8, 0.7, 0.4, 0.95, 0.6])
    assert OL.sample_by_keys(cls, keys, 20, 8, 0.25).tolist() == [1, 4, 2, 5, 6, 7, 9, 11]
    assert OL.sample_by_keys(cls, keys, 20, 64, 0.25).tolist() == [1, 4, 8, 0, 2, 5, 6, 7, 9, 10, 11]


def test_cascade_training_forward_matches_the_reference(golden_dir):
    """`oracle/losses.py::cascade_training_losses` against `tests/golden/cascade_training.npz`: the reference's own
    `DeticCascadeROIHeads.forward` run in TRAINING mode (detic_roi_heads.py:226-249, 88-147, 306-349) with three real
    `DeticFastRCNNOutputLayers` + `ZeroShotClassifier`s (`gen_golden_cascade_training.py`; detectron2's matcher / sampler / pooler are
    injected from the oracle's restatements there, so this pins the reference's OWN part: the stage chaining with the training-only
    non-empty filter, the re-labelling per stage, the loss terms and their names, loss_mask without gt_masks)."""
    import _inputs as I
    from oracle import model as M
    g = np.load(os.path.join(golden_dir, "cascade_training.npz"))
    feats, boxes, scores = I.cascade_case()
    sd = I.cascade_weights()
    zs = torch.from_numpy(np.load(os.path.join(golden_dir, "cascade.npz"))["zs_weight"])
    for k in range(3):
        sd[f"roi_heads.box_predictor.{k}.cls_score.zs_weight"] = zs
    gt, gc, keys = torch.from_numpy(g["gt_boxes"]), torch.from_numpy(g["gt_classes"]), torch.from_numpy(g["keys"])
    with torch.no_grad():
        losses, stages = OL.cascade_training_losses(feats, boxes, gt, gc, sd, M.OracleCfg(), I.CASCADE_HW, keys,
                                                    ious=tuple(float(v) for v in g["ious"]), batch=int(g["batch"]))
    names = [f"loss_{n}_stage{k}" for k in range(3) for n in ("cls", "box_reg")] + ["loss_mask"]
    assert sorted(losses) == sorted(names)
    for n in names:
        assert abs(float(losses[n]) - float(g[n])) <= 1e-5 * max(abs(float(g[n])), 1e-3), (n, float(losses[n]), float(g[n]))
    assert float(g["loss_box_reg_stage1"]) > 0 and float(g["loss_mask"]) == 0.0
    for k in range(3):
        np.testing.assert_allclose(stages[k]["boxes"].numpy(), g[f"stage{k}_boxes"], rtol=1e-5, atol=1e-4)
    assert np.array_equal(stages[0]["boxes"].numpy(), g["stage0_boxes"])                        # the sampled rows themselves
    assert np.array_equal(stages[0]["classes"].numpy(), g["sampled_classes"]) and int((g["sampled_classes"] < 20).sum()) == 8
    allb = torch.cat([boxes, gt])
    assert np.array_equal(allb[torch.from_numpy(g["sampled_rows"])].numpy(), g["stage0_boxes"]) and len(g["sampled_rows"]) == 32
