"""Oracle of the spatial feature memory (a3, a4, a16-a20 of SURVEY.md §8a).  TEST INFRASTRUCTURE ONLY.

Follows `Detic/detic/modeling/meta_arch/custom_rcnn.py:435-546,681-1042`.  The Detic-owned
functions here are pinned against outputs of the reference's own functions
(`tests/golden/gen_golden.py`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import model as M
from . import ops


def create_implicit_memory(memory: torch.Tensor, observations: torch.Tensor) -> torch.Tensor:
    """`custom_rcnn.py:762-774`: mem[obs>1] /= obs[obs>1] on a clone."""
    mem = memory.clone()
    sel = observations > 1
    mem[sel] = mem[sel] / observations.unsqueeze(1)[sel]
    return mem


def inference_with_proposals(proposals: Dict[str, torch.Tensor], zs_weight: torch.Tensor, thresh: float,
                             image_hw: Tuple[int, int], norm_temp: float = 50.0):
    """`custom_rcnn.py:825-882`.  Returns None or (boxes [K,4], feats [K,512], masks bool [K,H,W],
    kept proposal rows [K])."""
    feat = proposals["feat"]
    boxes = proposals["proposal_boxes"]
    masks = proposals["pred_masks"]
    ps = proposals["scores"]
    sel = torch.where(ps < 1)[0]
    ps, boxes, masks, feat = ps[sel], boxes[sel], masks[sel], feat[sel]
    feat = norm_temp * F.normalize(feat, p=2, dim=1)
    sc = torch.mm(feat, zs_weight).sigmoid()
    sc = (sc * ps[:, None]) ** 0.5
    _, det_scores, _, rows = ops.fast_rcnn_inference_single(boxes, sc, image_hw, thresh, 0.5, 100)
    if det_scores.numel() == 0:
        return None
    rows = torch.unique(rows)
    boxes = boxes[rows]
    feat = feat[rows]
    m = masks[rows].squeeze(1)
    pasted = ops.paste_masks(m, boxes, image_hw, 0.5)
    return boxes, feat, pasted, sel[rows]


def box_to_image_features(box_features: torch.Tensor, masks: torch.Tensor):
    """`custom_rcnn.py:884-901` (dense form, small sizes only): -> ([1,512,H,W] f32, bool [H,W])."""
    K, H, W = masks.shape
    image_features = torch.zeros((1, 512, H, W), dtype=torch.float32)
    observations = torch.zeros((1, 1, H, W), dtype=torch.float32)
    for i in range(K):
        mask = masks[i]
        image_features[:, :, mask] += box_features[i].reshape(1, 512, 1)
        observations[:, :, mask] += 1
    observed = (observations > 0).squeeze(0).squeeze(0)
    image_features[:, :, observed] = image_features[:, :, observed] / observations[:, :, observed]
    return image_features, observed


def project_image_features(image_features: torch.Tensor, observed: torch.Tensor, proj: torch.Tensor, n_cells: int):
    """`custom_rcnn.py:903-936` without the dense one-hot: every 8th observed pixel (row-major),
    grouped by cell, mean per cell.  -> (mean [U,512] in ascending cell order, observed_mem bool [N])."""
    feats = image_features[:, :, observed].squeeze(0).permute(1, 0).reshape(-1, 512)
    pr = proj[observed]
    pr = pr[::8]
    feats = feats[::8]
    sums = torch.zeros((n_cells, 512), dtype=torch.float32)
    cnt = torch.zeros((n_cells,), dtype=torch.float32)
    sums.index_add_(0, pr, feats.to(torch.float32))
    cnt.index_add_(0, pr, torch.ones_like(pr, dtype=torch.float32))
    observed_mem = cnt > 0
    mean = sums[observed_mem] / cnt[observed_mem].unsqueeze(1)
    return mean, observed_mem


def memory_write_sparse(box_features: torch.Tensor, masks: torch.Tensor, proj: torch.Tensor, n_cells: int):
    """a17+a18 fused without the [1,512,H,W] temporary; identical arithmetic order: per sampled pixel the
    instance features are added in instance order and divided by the cover count, per cell the sampled
    pixels are added in row-major order and divided by their number."""
    K, H, W = masks.shape
    count = masks.sum(dim=0).to(torch.float32)
    observed = count > 0
    pix = torch.nonzero(observed.reshape(-1)).squeeze(1)[::8]
    acc = torch.zeros((pix.numel(), 512), dtype=torch.float32)
    mflat = masks.reshape(K, -1)[:, pix]
    for i in range(K):
        acc += mflat[i].to(torch.float32)[:, None] * box_features[i][None, :]
    acc = acc / count.reshape(-1)[pix][:, None]
    pr = proj.reshape(-1)[pix]
    sums = torch.zeros((n_cells, 512), dtype=torch.float32)
    cnt = torch.zeros((n_cells,), dtype=torch.float32)
    sums.index_add_(0, pr, acc)
    cnt.index_add_(0, pr, torch.ones_like(pr, dtype=torch.float32))
    observed_mem = cnt > 0
    mean = sums[observed_mem] / cnt[observed_mem].unsqueeze(1)
    return mean, observed_mem


def semmap_labels(semmap_features: torch.Tensor, observation_count: torch.Tensor, zs_weight: torch.Tensor,
                  thresh: float) -> torch.Tensor:
    """a20: `custom_rcnn.py:745-756,938-1017` on flat [N,512]/[N] state -> int32 [N] labels (-1 below thresh)."""
    inten = semmap_features.abs().mean(dim=1)
    sel = observation_count > 1
    inten = torch.where(sel, inten / observation_count, inten)
    inten = (inten - inten.min()) / (inten.max() - inten.min())
    nf = 50.0 * F.normalize(semmap_features, p=2, dim=1)
    sc = torch.mm(nf, zs_weight)[:, :20].softmax(dim=1)
    idx = sc.argmax(dim=1).to(torch.int32)
    idx = torch.where(inten < thresh, torch.full_like(idx, -1), idx)
    return idx


class RecurrentOracle:
    """Eval branch of `CustomRCNNRecurrent.forward` (`custom_rcnn.py:435-546`) as a CPU state machine."""

    def __init__(self, sd, cfg: Optional[M.OracleCfg] = None):
        self.sd = sd
        self.cfg = cfg or M.OracleCfg()
        self.zs_weight = sd["roi_heads.box_predictor.0.cls_score.zs_weight"]
        self.implicit_memory = None
        self.observations = None
        self.semmap_features = None
        self.observation_count = None
        self._snap_mem = None
        self._snap_obs = None
        self.last = {}
        self.timings = None          # dict: seconds per stage are accumulated into it (bench.py's cpu_baseline leg)

    def reset(self, n_cells: int):
        self.semmap_features = None
        self.observation_count = None
        self.implicit_memory = torch.zeros((n_cells, 512), dtype=torch.float32)
        self.observations = torch.zeros((n_cells,), dtype=torch.float32)

    def __call__(self, batched_inputs: List[List[dict]]):
        out = []
        for seq in batched_inputs:
            for i, frame in enumerate(seq):
                out.append(self.step(frame, i, seq))
        return out

    def step(self, frame: dict, i: int = 0, seq: Optional[list] = None):
        cfg = self.cfg
        n_cells = int((seq[0] if seq else frame)["memory"].shape[0])
        if frame["memory_reset"]:
            self.reset(n_cells)
        if i == 0 and cfg.test_type == "longterm":
            self._snap_mem, self._snap_obs = self.implicit_memory, self.observations
        if cfg.test_type in ("default", "episodic"):
            self._snap_mem, self._snap_obs = self.implicit_memory, self.observations
        proj = torch.as_tensor(np.asarray(frame["proj_indices"])).to(torch.long)
        if proj.dim() == 3:
            proj = proj.squeeze(2)
        image = frame["image"]
        H, W = image.shape[1:]
        import time as _t
        t0 = _t.perf_counter()
        mem_f16 = None
        if cfg.memory_type == "implicit_memory":
            mem = create_implicit_memory(self._snap_mem, self._snap_obs)
            mem_f16 = mem.to(torch.half)                    # preprocess_spatial_memory :1036
        t1 = _t.perf_counter()
        kw = {} if self.timings is None else {"timings": self.timings}
        proposals, result = M.inference(self.sd, cfg, image, mem_f16, proj, (frame.get("height", H), frame.get("width", W)), **kw)
        t2 = _t.perf_counter()
        self.update_implicit_memory(proposals, proj, n_cells, (H, W))
        if self.timings is not None:
            self.timings["create_implicit_memory + fp16 cast"] = self.timings.get("create_implicit_memory + fp16 cast", 0.0) + (t1 - t0)
            self.timings["memory write (update_implicit_memory)"] = (self.timings.get("memory write (update_implicit_memory)", 0.0)
                                                                      + (_t.perf_counter() - t2))
        return {"instances": result, "proposals": proposals}

    def update_implicit_memory(self, proposals, proj, n_cells, image_hw):
        """`custom_rcnn.py:681-760` (runs for every MEMORY_TYPE)."""
        res = inference_with_proposals(proposals, self.zs_weight, self.cfg.memory_cls_score_thresh, image_hw,
                                       self.cfg.norm_temp)
        self.last = {"K": 0}
        if res is None:
            return
        boxes, feats, masks, rows = res
        mean, observed_mem = memory_write_sparse(feats, masks, proj, n_cells)
        update = torch.zeros((n_cells, 512), dtype=torch.float32)
        update[observed_mem] = mean
        obs_update = torch.zeros((n_cells,), dtype=torch.float32)
        obs_update[torch.unique(proj)] = 1
        if self.semmap_features is None:
            self.semmap_features = update
            self.observation_count = obs_update
        else:
            self.semmap_features = self.semmap_features + update
            self.observation_count = self.observation_count + obs_update
        self.implicit_memory = self.semmap_features
        self.observations = self.observation_count
        self.last = {"K": int(boxes.shape[0]), "rows": rows, "masks": masks, "feats": feats, "boxes": boxes,
                     "masks28": proposals["pred_masks"][rows].squeeze(1), "observed_mem": observed_mem, "mean": mean}
