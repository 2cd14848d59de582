"""Training forward + backward of the proposal half of `forward_model` (`custom_rcnn.py:584-679` up to the proposal
losses): image -> backbone with the memory read fused in -> CenterNet head -> target assignment -> `CenterNet.losses` -> gradients of
every parameter upstream (head tower + GroupNorm + output convs + level scales, FPN, map_merge projections, ResNet-50 trunk), all on
the HIP kernels.  The ROI heads' half (proposal matching, cascade / mask losses) is not part of it.

Not the inference hot path: the head here keeps each layer's activations and runs its 5-channel output conv in a 32-channel tile
(the weight-gradient kernel's tile width); the arithmetic per layer is the hot path's.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops
from .backward import BackboneBackward


class ProposalTraining:
    def __init__(self, model, sd: Dict[str, torch.Tensor]):
        """`model`: the built `CustomRCNNRecurrent`; `sd`: its state dict (fp32 masters of the parameters the step differentiates)."""
        self.model = model
        self.dev = model.device
        self.pg = model.proposal_generator
        c = model.cfg.MODEL.CENTERNET
        self.loss_cfg = dict(alpha=float(c.HM_FOCAL_ALPHA), beta=float(c.HM_FOCAL_BETA), gamma=float(c.LOSS_GAMMA),
                             sigmoid_clamp=float(c.SIGMOID_CLAMP), ignore_high_fp=float(c.IGNORE_HIGH_FP), pos_weight=float(c.POS_WEIGHT),
                             neg_weight=float(c.NEG_WEIGHT), reg_weight=float(c.REG_WEIGHT))
        if str(c.LOC_LOSS_TYPE) != "giou" or not bool(c.NOT_NORM_REG) or bool(c.MORE_POS) or bool(c.NO_REDUCE):
            raise NotImplementedError("proposal losses: LOC_LOSS_TYPE giou, NOT_NORM_REG, no MORE_POS / NO_REDUCE (the recurrent yaml)")
        self.target_cfg = dict(strides=list(c.FPN_STRIDES), sizes_of_interest=[tuple(x) for x in c.SOI],
                               hm_min_overlap=float(c.HM_MIN_OVERLAP), min_radius=float(c.MIN_RADIUS))
        self.bb = BackboneBackward(model.backbone, [sd[f"backbone.map_merge_projection{i}.weight"] for i in (1, 2, 3)])
        h = "proposal_generator.centernet_head"
        w = torch.zeros((32, 256, 3, 3))
        b = torch.zeros((32,))
        w[:5] = torch.cat([sd[f"{h}.agn_hm.weight"], sd[f"{h}.bbox_pred.weight"]], dim=0).float()
        b[:5] = torch.cat([sd[f"{h}.agn_hm.bias"], sd[f"{h}.bbox_pred.bias"]], dim=0).float()
        self.out32 = ops.Conv(w, b, pad=1, device=self.dev, name="agn_hm+bbox_pred")
        self._bw: Dict[int, ops.ConvBackward] = {}
        self._loss = {}
        self.last = None

    # ---- helpers -------------------------------------------------------------------------------------------------------------------
    def _conv_bwd(self, conv: ops.Conv, xin: torch.Tensor, gout: torch.Tensor, shapes, off):
        """Backward of a level-shared conv over the pyramid row list: per level on that level's grid, dW / db summed over the levels."""
        if id(conv) not in self._bw:
            self._bw[id(conv)] = ops.ConvBackward(conv)
        bwd = self._bw[id(conv)]
        dw = db = None
        dxs = []
        for l, (h, w) in enumerate(shapes):
            o = bwd(xin[off[l]:off[l + 1]].view(1, h, w, conv.Cin), None, gout[off[l]:off[l + 1]].view(1, h, w, conv.Cout))
            dw = o["dw"] if dw is None else dw.add_(o["dw"])
            db = o["db"] if db is None else db.add_(o["db"])
            dxs.append(o["dx"].reshape(-1, conv.Cin))
        return torch.cat(dxs), dw, db

    # ---- one step ------------------------------------------------------------------------------------------------------------------
    def forward_backward(self, image_u8: torch.Tensor, gt_boxes: torch.Tensor, memory=None, world_size: int = 1, reduce_counts=None):
        """image_u8 [3,H,W] on the device, gt_boxes [N,4] fp32 on the device, `memory = (memory_f16, proj_indices)` or None ->
        (losses {name: float tensor on the device}, grads {layer or parameter name: tensors}).

        `reduce_counts(counts int32 [2]) -> counts` sums the positives / regression rows over the ranks (the reference's
        `reduce_sum`, centernet.py:263-265,293); both are then divided by `world_size`."""
        m, pg, dev = self.model, self.pg, self.dev
        x4, Hp, Wp = ops.preprocess_image(image_u8, m.pixel_mean, m.pixel_std)
        P, saved = self.bb.forward(x4, Hp, Wp, memory=memory)
        shapes = [(p.shape[1], p.shape[2]) for p in P]
        off = [0]
        for (h, w) in shapes:
            off.append(off[-1] + h * w)
        feats = torch.cat([p.reshape(-1, 256) for p in P])
        # CenterNet head (centernet_head.py:141-161), every layer's input / pre-norm / output kept
        keep, x = [], feats
        for (conv, gamma, beta) in pg.tower:
            c = conv(x, 1, 0, 0, levels=(off, shapes))
            st = ops.groupnorm_workspace(off, dev)
            y = ops.groupnorm_relu(c, gamma, beta, off, 256, st)
            keep.append((x, c, st, y))
            x = y
        head = self.out32(x, 1, 0, 0, levels=(off, shapes))                                   # [P, 32]: cols 0..4 are the head's
        self.last = dict(keep=keep, head=head, shapes=shapes, off=off, saved=saved)                        # for inspection (tests)
        # targets (centernet.py:342-479) and losses (:241-318)
        heat, reg_t, pos, counts = ops.centernet_targets(gt_boxes, shapes, **self.target_cfg)
        local = counts.cpu().tolist()                                                         # the reference's `.item()` (:264,293)
        total = reduce_counts(counts).cpu().tolist() if reduce_counts is not None else local
        key = tuple(shapes)
        if key not in self._loss:
            self._loss[key] = ops.CenterNetLoss(off, pg.scales, dev, head_stride=32, **self.loss_cfg)
        losses_t, d_head = self._loss[key](head, heat, reg_t, pos[:local[0]], max(total[0] / world_size, 1.0),
                                           max(total[1] / world_size, 1.0))
        losses = {"loss_centernet_loc": losses_t[0], "loss_centernet_agn_pos": losses_t[1], "loss_centernet_agn_neg": losses_t[2]}
        # ---- backward
        grads: Dict[str, tuple] = {}
        # the levels' Scale parameters (centernet_head.py:153-155): d scale_l = sum over the level of d reg * raw = d raw * raw / scale_l
        prod = (d_head[:, 1:5] * head[:, 1:5]).sum(dim=1)
        grads["scales"] = torch.stack([prod[off[l]:off[l + 1]].sum() / pg.scales[l] for l in range(len(shapes))])
        gx, dw, db = self._conv_bwd(self.out32, x, d_head, shapes, off)
        grads["agn_hm"] = (dw[0:1], db[0:1])
        grads["bbox_pred"] = (dw[1:5], db[1:5])
        for i in reversed(range(len(pg.tower))):
            conv, gamma, beta = pg.tower[i]
            xin, c, st, y = keep[i]
            dc, dgamma, dbeta = ops.groupnorm_relu_backward(c, y, gx, gamma, off, 256, st)
            gx, dw, db = self._conv_bwd(conv, xin, dc, shapes, off)
            grads[conv.name] = (dw, db)
            grads[conv.name + ".norm"] = (dgamma, dbeta)
        dP = [gx[off[l]:off[l + 1]].view(1, shapes[l][0], shapes[l][1], 256) for l in range(len(shapes))]
        bgrads, _ = self.bb.backward(saved, dP)
        grads.update(bgrads)
        return losses, grads
