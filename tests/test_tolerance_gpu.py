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
    from embodied_object_detection_amd import build_model
    from embodied_object_detection_amd.data.synthetic import SyntheticSequence
    seq = SyntheticSequence(3, H=H, W=W, n_frames=n_frames, map_w=grid, map_h=grid, cell=0.5 if grid < 200 else 0.2)
    frames = [seq.frame(i) for i in range(n_frames)]
    model = build_model(_cfg(), synthetic_sd)
    oracle = OM.RecurrentOracle(synthetic_sd, M.OracleCfg(memory_cls_score_thresh=0.3, map_feature_weight=5.0))
    outs = model([frames])                                     # one episode through the boundary (Instances materialised)
    report = []
    for i, f in enumerate(frames):
        ref = oracle.step(f, i, frames)["instances"]
        r = compare(ref, outs[i]["instances"])
        r["frame"] = i
        # the memory the NEXT frame reads
        r["memory_max_abs_err"] = float((model.implicit_memory.cpu() - oracle.implicit_memory).abs().max()) if i == n_frames - 1 else None
        report.append(r)
        print(f"[parity {H}x{W} frame {i}] {r}")
    assert torch.equal(model.observations.cpu(), oracle.observations)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"parity_report_{H}x{W}.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    for r in report:
        assert r["matched"] >= 0.98 * r["n_ref"] and abs(r["n_ref"] - r["n_got"]) <= max(3, 0.02 * r["n_ref"]), r
        assert r["max_abs_dscore"] < TOL, r
        assert r["max_abs_dbox_px"] < TOL, r
