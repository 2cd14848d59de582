"""COCO bbox AP / AP50 as the reference's eval uses it (the record buffer / collective live in engine/eval_loop.py).

Reference call sites: `Detic/train_mp3d.py:246,301-358` (`COCOEvaluator` -> pycocotools `COCOeval`; neither package
is vendored: semantics per SURVEY.md Appendix A14).  Host-side numpy: this is evaluation bookkeeping, not the hot path.

Multi-GPU: scenes are sharded by scene id, every rank writes its detections / GT boxes into its own slice of
a `[world, rows, 8]` fp32 buffer, `all_reduce(SUM)` (RCCL over xGMI; slices are disjoint so SUM == all-gather) and rank
0 runs the global per-class accumulation.  COCO matching is per (image, class), hence unaffected by the sharding.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

KIND_DET, KIND_GT = 1.0, 2.0
IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)


def box_iou_xyxy(d: np.ndarray, g: np.ndarray) -> np.ndarray:
    """IoU matrix [len(d), len(g)] without the +1 convention."""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    x1 = np.maximum(d[:, None, 0], g[None, :, 0])
    y1 = np.maximum(d[:, None, 1], g[None, :, 1])
    x2 = np.minimum(d[:, None, 2], g[None, :, 2])
    y2 = np.minimum(d[:, None, 3], g[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    ad = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1])
    ag = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    union = ad[:, None] + ag[None, :] - inter
    return np.where(union > 0, inter / np.maximum(union, 1e-30), 0.0)


def coco_eval(dets: Dict[int, dict], gts: Dict[int, dict], num_classes: int, max_dets: int = 100,
              image_ids: Optional[List[int]] = None) -> Dict[str, float]:
    """dets[img] = {'boxes' [D,4] xyxy, 'scores' [D], 'classes' [D]}; gts[img] = {'boxes' [G,4], 'classes' [G]}.
    Returns AP (IoU .50:.95), AP50, AP75 in percent; categories without GT are excluded (precision -1 in COCOeval)."""
    imgs = sorted(set(gts.keys()) | set(dets.keys())) if image_ids is None else list(image_ids)
    T = len(IOU_THRS)
    ap_per_cat = np.full((T, num_classes), -1.0)
    for c in range(num_classes):
        scores_all, match_all = [], []
        npig = 0
        for img in imgs:
            g = gts.get(img)
            gb = g["boxes"][g["classes"] == c] if g is not None and len(g["boxes"]) else np.zeros((0, 4))
            d = dets.get(img)
            if d is not None and len(d["boxes"]):
                sel = d["classes"] == c
                db, ds = d["boxes"][sel], d["scores"][sel]
            else:
                db, ds = np.zeros((0, 4)), np.zeros((0,))
            npig += len(gb)
            if len(db) == 0:
                continue
            order = np.argsort(-ds, kind="mergesort")[:max_dets]
            db, ds = db[order], ds[order]
            ious = box_iou_xyxy(db, gb)
            dtm = np.zeros((T, len(db)), dtype=bool)
            for ti, t in enumerate(IOU_THRS):
                gtm = np.zeros(len(gb), dtype=bool)
                for di in range(len(db)):
                    best, m = min(t, 1 - 1e-10), -1
                    for gi in range(len(gb)):
                        if gtm[gi] or ious[di, gi] < best:
                            continue
                        best, m = ious[di, gi], gi
                    if m >= 0:
                        gtm[m] = True
                        dtm[ti, di] = True
            scores_all.append(ds)
            match_all.append(dtm)
        if npig == 0:
            continue
        if not scores_all:
            ap_per_cat[:, c] = 0.0
            continue
        sc = np.concatenate(scores_all)
        tpm = np.concatenate(match_all, axis=1)
        order = np.argsort(-sc, kind="mergesort")
        tpm = tpm[:, order]
        for ti in range(T):
            tp = np.cumsum(tpm[ti]).astype(np.float64)
            fp = np.cumsum(~tpm[ti]).astype(np.float64)
            rc = tp / npig
            pr = tp / (fp + tp + np.spacing(1))
            pr = pr.tolist()
            for i in range(len(pr) - 1, 0, -1):
                if pr[i] > pr[i - 1]:
                    pr[i - 1] = pr[i]
            inds = np.searchsorted(rc, REC_THRS, side="left")
            q = np.zeros(len(REC_THRS))
            for ri, pi in enumerate(inds):
                if pi < len(pr):
                    q[ri] = pr[pi]
            ap_per_cat[ti, c] = q.mean()

    def mean_valid(a):
        v = a[a > -1]
        return float(v.mean() * 100) if v.size else float("nan")

    return {"AP": mean_valid(ap_per_cat), "AP50": mean_valid(ap_per_cat[0]), "AP75": mean_valid(ap_per_cat[5]),
            "num_images": len(imgs)}
