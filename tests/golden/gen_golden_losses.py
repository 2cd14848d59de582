#!/usr/bin/env python
"""Golden vectors of the proposal generator's training losses  --  runs ONLY in the development container (needs /root/reference).

Loads the reference's OWN loss modules from where they lie (`Detic/third_party/CenterNet2/centernet/modeling/layers/
heatmap_focal_loss.py`, `iou_loss.py`: plain torch, no detectron2) and evaluates `binary_heatmap_focal_loss_jit` and
`IOULoss('giou')` exactly as `CenterNet.losses` calls them (`.../dense_heads/centernet.py:283-313`) on seeded inputs of a five-level
pyramid, with autograd's gradients with respect to the agnostic logits and the regression predictions.  Stores inputs, the three
losses and the gradients in `tests/golden/centernet_loss.npz`.  No reference source or bytecode is copied.

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


if __name__ == "__main__":
    main()
