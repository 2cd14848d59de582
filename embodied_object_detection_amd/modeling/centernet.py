"""CenterNet proposal generator (ONLY_PROPOSAL + WITH_AGN_HM) on the HIP kernels.

Mirrors `CenterNetHead.forward` (`Detic/third_party/CenterNet2/centernet/modeling/dense_heads/centernet_head.py:141-161`)
and `CenterNet.inference/predict_instances/predict_single_level/nms_and_topK`
(`.../dense_heads/centernet.py:603-745`).  The tower weights are shared across the five levels, so every tower
layer runs on the pyramid's concatenated row list; `agn_hm` (1 channel) and `bbox_pred` (4 channels) are merged into
one 5-channel 3x3 conv; Scale / ReLU / stride multiply, sigmoid, threshold, top-k, NMS and the post-NMS cut all run
on device with a device-side proposal count (the reference syncs to the host for kthvalue, centernet.py:735-738).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .. import ops
from ..registry import PROPOSAL_GENERATOR_REGISTRY


@PROPOSAL_GENERATOR_REGISTRY.register()
class CenterNet:
    def __init__(self, cfg, sd: Dict[str, torch.Tensor], device):
        c = cfg.MODEL.CENTERNET
        if not (c.ONLY_PROPOSAL and c.WITH_AGN_HM):
            raise NotImplementedError("hot path covers ONLY_PROPOSAL + WITH_AGN_HM (Base-...recurrent.yaml:42-43)")
        if c.NORM != "GN" or c.NUM_BOX_CONVS != 4 or c.NUM_SHARE_CONVS != 0 or c.USE_DEFORMABLE or c.CENTER_NMS or c.NOT_NMS:
            raise NotImplementedError("unsupported MODEL.CENTERNET variant for the recurrent path")
        self.device = device
        self.strides = list(c.FPN_STRIDES)
        self.score_thresh = float(c.INFERENCE_TH)
        self.pre_nms_topk = int(c.PRE_NMS_TOPK_TEST)
        self.post_nms_topk = int(c.POST_NMS_TOPK_TEST)
        self.nms_thresh = float(c.NMS_TH_TEST)
        # capacity for the '>= kth' tie rule (centernet.py:739): ties beyond it would be dropped and flagged
        self.cap = (self.post_nms_topk + 64 + 31) // 32 * 32
        h = "proposal_generator.centernet_head"
        self.tower = []
        for i in range(4):
            conv = ops.Conv(sd[f"{h}.bbox_tower.{3 * i}.weight"], sd[f"{h}.bbox_tower.{3 * i}.bias"], pad=1, device=device,
                            name=f"bbox_tower.{3 * i}")
            gamma = sd[f"{h}.bbox_tower.{3 * i + 1}.weight"].to(device)
            beta = sd[f"{h}.bbox_tower.{3 * i + 1}.bias"].to(device)
            self.tower.append((conv, gamma, beta))
        w = torch.cat([sd[f"{h}.agn_hm.weight"], sd[f"{h}.bbox_pred.weight"]], dim=0)   # [5,256,3,3]: row 0 agn_hm, 1..4 bbox_pred
        b = torch.cat([sd[f"{h}.agn_hm.bias"], sd[f"{h}.bbox_pred.bias"]], dim=0)
        self.out_conv = ops.Conv(w, b, pad=1, device=device, name="agn_hm+bbox_pred")
        self.scales = [float(sd[f"{h}.scales.{l}.scale"].item()) for l in range(5)]
        self._plans = {}
        self.fuse_gn_stats = True
        # EodConvDesc.force_tile of the 5-channel output conv: 6 / 7 = K over 4 / 8 waves, no slab reduce launch -- measured in the frame
        # (tools/knob_ab.py, same call): 288.6 frames/s with the planner's slabs + reduce, 287.7 / 287.2 with 6 / 7: stays 0
        self.out_conv_tile = 0

    def _plan(self, shapes: List[Tuple[int, int]], off: List[int]):
        key = tuple(shapes)
        if key not in self._plans:
            P = off[-1]
            a = torch.empty((P, 256), dtype=torch.float32, device=self.device)
            b = torch.empty((P, 256), dtype=torch.float32, device=self.device)
            head = torch.empty((P, 5), dtype=torch.float32, device=self.device)
            dec = ops.ProposalDecoder(shapes, self.strides, self.scales, self.score_thresh, self.pre_nms_topk, self.post_nms_topk,
                                      self.nms_thresh, self.cap, self.device, head_stride=5)
            self._plans[key] = (a, b, head, dec, ops.groupnorm_workspace(off, self.device))
        return self._plans[key]

    def _per_level(self, conv, src: torch.Tensor, dst: torch.Tensor, shapes, off, cout: int, gn_stats=None, force_tile: int = 0):
        # one launch over the whole pyramid (weights are shared across levels, centernet_head.py:144-160)
        conv(src, 1, 0, 0, out=dst, levels=(off, shapes), gn_stats=gn_stats, force_tile=force_tile)

    def forward(self, feats: torch.Tensor, shapes, off):
        """feats [P_total,256] -> (boxes [cap,4], scores [cap], count [1]) device buffers, sorted by score."""
        a, b, head, dec, gn_ws = self._plan(shapes, off)
        src = feats
        for (conv, gamma, beta) in self.tower:
            # `fuse_gn_stats`: the conv's slab reduce also writes GroupNorm's partial sums (one launch less per tower layer on the
            # frame's critical chain: +0.6 % frames/s in a same-call A/B, tools/knob_ab.py).  The reduce keeps its coalesced float4
            # mapping and accumulates the sums per thread in double (a first version that walked one channel per thread took 44 us
            # against 12 + 11 for the separate launches and was off).
            self._per_level(conv, src, a, shapes, off, 256, gn_stats=gn_ws if self.fuse_gn_stats else None)
            ops.groupnorm_relu(a, gamma, beta, off, 256, gn_ws, out=b, partial_ready=conv.gn_fused)   # stream order: `a` is free again
            src = b
        self._per_level(self.out_conv, src, head, shapes, off, 5, force_tile=self.out_conv_tile)
        return dec(head)
