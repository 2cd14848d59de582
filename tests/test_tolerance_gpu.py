"""north_star tolerance, measured and asserted in ABSOLUTE units: boxes (pixels) and scores of the HIP model's detections against
the CPU oracle on identical frames, through the boundary `model([[frame]])`, at 128x160, 480x640 (config A) and 640x640 (config B).

A detection is *matched* when the oracle has one of the same class with IoU > 0.99; unmatched ones are selection flips (top-k / NMS /
threshold decisions on near-ties between two fp32 implementations) and are bounded separately.  For every matched detection the
absolute coordinate and score differences must be below 1e-3.  The measured maxima are printed and written to
gpurun_out/parity_report_<size>.json so that DESIGN.md quotes numbers, not bounds."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import memory as OM
from oracle import model as M
from oracle import ops as OO

TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 on box coords/scores"


def _cfg():
    from embodied_object_detection_amd import setup_cfg
    return setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5,
                            "MODEL.MEMORY_CLS_SCORE_THRESH", 0.3])


def compare(ref, out):
    rb, rs, rc = ref["pred_boxes"], ref["scores"], ref["pred_classes"]
    gb, gs, gc = out.pred_boxes.tensor.cpu(), out.scores.cpu(), out.pred_classes.cpu()
    matched, dbox, dscore = 0, 0.0, 0.0
    for b, s, c in zip(rb, rs, rc):
        cand = (gc == c).nonzero().squeeze(1)
        if not cand.numel():
            continue
        iou = OO.iou_one_to_many(b, gb[cand])
        j = int(iou.argmax())
        if float(iou[j]) > 0.99:
            matched += 1
            dbox = max(dbox, float((gb[cand[j]] - b).abs().max()))
            dscore = max(dscore, float((gs[cand[j]] - s).abs()))
    return dict(n_ref=int(rb.shape[0]), n_got=int(gb.shape[0]), matched=matched, max_abs_dbox_px=dbox, max_abs_dscore=dscore)


@pytest.mark.parametrize("H,W,grid,n_frames", [(128, 160, 24, 4), (480, 640, 60, 2), (640, 640, 200, 2)])
def test_absolute_tolerance_through_the_boundary(synthetic_sd, H, W, grid, n_frames):
    """"On identical frames" includes the recurrent state: before every frame the HIP model's memory is set to the oracle's
    (teacher forcing), so each frame measures ONE pass of the path.  (Free-running, the path amplifies last-bit differences through
    its discrete steps -- a mask pixel flipping at the 0.5 threshold shifts the phase of the every-8th-observed-pixel rule,
    custom_rcnn.py:913-914, and with it which cells are written; that drift is reported, not asserted.)"""
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    seq = SyntheticSequence(3, H=H, W=W, n_frames=n_frames, map_w=grid, map_h=grid, cell=0.5 if grid < 200 else 0.2)
    frames = [seq.frame(i) for i in range(n_frames)]
    model = build_model(_cfg(), synthetic_sd)
    free = build_model(_cfg(), synthetic_sd)
    oracle = OM.RecurrentOracle(synthetic_sd, M.OracleCfg(memory_cls_score_thresh=0.3, map_feature_weight=5.0))
    free_outs = free([frames])                                 # one episode through the boundary, free running
    report = []
    for i, f in enumerate(frames):
        if i > 0:
            model.implicit_memory.copy_(oracle.implicit_memory.to(model.device))
            model.observations.copy_(oracle.observations.to(model.device))
            model.invalidate_memory_snapshot()
        g = dict(f)
        g["memory_reset"] = f["memory_reset"] and i == 0
        out = model([[g]])[0]["instances"]                     # through the boundary, Instances materialised
        mem_before = None if oracle.implicit_memory is None else oracle.implicit_memory.clone()
        ref = oracle.step(f, i, frames)["instances"]
        r = compare(ref, out)
        r["frame"] = i
        r["observations_exact"] = bool(torch.equal(model.observations.cpu(), oracle.observations))
        got_mem, ref_mem = model.implicit_memory.cpu(), oracle.implicit_memory
        base = torch.zeros_like(ref_mem) if mem_before is None else mem_before
        r["written_cells_identical"] = bool(torch.equal((got_mem != base).any(dim=1), (ref_mem != base).any(dim=1)))
        r["memory_max_abs_err"] = float((got_mem - ref_mem).abs().max())
        r["free_running"] = compare(ref, free_outs[i]["instances"])
        report.append(r)
        print(f"[parity {H}x{W} frame {i}] {r}")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"parity_report_{H}x{W}.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    for r in report:
        assert r["observations_exact"], r
        assert r["matched"] >= 0.98 * r["n_ref"] and abs(r["n_ref"] - r["n_got"]) <= max(3, 0.02 * r["n_ref"]), r
        assert r["max_abs_dscore"] < TOL, r
        assert r["max_abs_dbox_px"] < TOL, r
