"""Backward of the backbone (training slices): the chain of the per-layer HIP backward kernels over the
ResNet-50 trunk (`Detic/detic/modeling/backbone/timm.py:277-299`), the FPN top-down pass (`timm.py:118-136`) and P6 / P7
(`timm.py:347-364`), driven by the gradients of the five pyramid levels the heads hand back.

Not part of the inference hot path: the forward here launches exactly the layers of `CustomRecurrentFPN.top_down` /
`fuse_memory_and_top` but keeps every activation the backward reads.  Gradients come back per `ops.Conv` (dW in the layer's
packed [Cout, KH*KW*Cin] layout, db) -- for the trunk these are the gradients of the FrozenBatchNorm-FOLDED weights: the raw
conv weight's gradient is dW * gamma / sqrt(var + eps) per output channel and db is the gradient of the norm's bias.  The 7x7
stem (4-channel tap layout) gets its weight gradient from a kernel of its own and no input gradient (its input is the image).
Residual sums between the kernels are plain device adds.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import _lib, ops
from .backbone import CustomRecurrentFPN


class BackboneBackward:
    def __init__(self, backbone: CustomRecurrentFPN, merge_weights: Optional[List[torch.Tensor]] = None, side_stream: bool = False):
        """`merge_weights`: the three `backbone.map_merge_projection{1,2,3}.weight` tensors ([256,512,1,1] fp32 masters), needed
        only when the forward runs with a memory.  `side_stream`: the layers' weight-gradient launches go to a second stream
        (`ops.ConvBackward`); the caller joins it (`ops.ConvBackward.join`) before it reads a dW / db."""
        self.side = bool(side_stream)
        if backbone.feat_fusion not in ("sum", "image_only"):
            raise ValueError("the backbone's backward covers MAP_FEAT_FUSION sum / image_only (mem_only has no image gradient)")
        self.bb = backbone
        self.lib = _lib.load()
        self._bw: Dict[int, ops.ConvBackward] = {}
        self.merge_weights = merge_weights
        self._merge_bw = None

    def _b(self, conv: ops.Conv) -> ops.ConvBackward:
        if id(conv) not in self._bw:
            self._bw[id(conv)] = ops.ConvBackward(conv, side_stream=self.side)
        return self._bw[id(conv)]

    # ---- forward that keeps its activations -------------------------------------------------------------------------------
    # Two halves: the TRUNK half (ResNet-50 + FPN laterals / top-down / output convs -> P3..P5 before the fusion) does not depend
    # on the memory and runs for N images in one launch per layer; the TAIL half (memory read + sum fusion, P6 / P7) is per scene.
    # `forward` = trunk + tail for the single-frame step; a batch of frames shares one trunk pass (`ProposalTraining.
    # forward_backward_batch`).
    def forward_trunk(self, x4: torch.Tensor, H: int, W: int, N: int = 1):
        """-> ([p3, p4, p5] as [N,h,w,256] before the memory fusion, keep)."""
        bb = self.bb
        keep: dict = {}
        c = bb.bottom_up.forward(x4, H, W, N, keep=keep)
        (c5, h5, w5), (c4, h4, w4), (c3, h3, w3) = c["layer5"], c["layer4"], c["layer3"]
        # N > 1: planned like a single image (`plan_rows`: the same split-K walk), so every image's levels are bitwise those of its
        # own N = 1 pass -- as the trunk's layers are (`ResNet50Trunk.forward`) -- and the selections behind them cannot flip
        pr = (lambda h, w: h * w) if N > 1 else (lambda h, w: 0)
        lat5 = bb.lateral[5](c5, N, h5, w5, plan_rows=pr(h5, w5))
        p5 = bb.output[5](lat5, N, h5, w5, plan_rows=pr(h5, w5))
        lat4 = bb.lateral[4](c4, N, h4, w4, res=lat5, res_mode=2, plan_rows=pr(h4, w4))
        p4 = bb.output[4](lat4, N, h4, w4, plan_rows=pr(h4, w4))
        lat3 = bb.lateral[3](c3, N, h3, w3, res=lat4, res_mode=2, plan_rows=pr(h3, w3))
        p3 = bb.output[3](lat3, N, h3, w3, plan_rows=pr(h3, w3))
        h6, w6 = bb.p6.out_hw(h5, w5)
        keep["fpn"] = dict(c=(c3, c4, c5), lat=(lat3, lat4, lat5), hw=((h3, w3), (h4, w4), (h5, w5), (h6, w6)))
        keep["N"], keep["x4"], keep["HW"] = N, x4, (H, W)
        return [p3, p4, p5], keep

    def forward_tail(self, p345: List[torch.Tensor], H: int, W: int, memory=None):
        """[p3, p4, p5] of n images ([n,h,w,256]; n == 1 with a memory) -> ([P3..P7], tail): the memory read + sum fusion into
        P3..P5 as the hot path runs it (`eod_memory_gather_pool` + `eod_memory_project_fuse`; the inputs are not written to), then
        P6 / P7.  `memory = (memory_f16, proj)` or None = image only."""
        bb = self.bb
        P = list(p345)
        n = int(P[0].shape[0])
        h5, w5 = int(P[2].shape[1]), int(P[2].shape[2])
        pooled = None
        if memory is not None and bb.feat_fusion == "sum":
            if n != 1:
                raise ValueError("the memory fusion is per scene: N == 1")
            rows = torch.cat([p.reshape(-1, 256) for p in P])
            pooled = ops.memory_gather_pool(memory[0], memory[1], H, W, torch_order=bb.pool_in_torch_order)
            bb.merge(pooled, rows, H, W, bb.map_feature_weight, bb.feat_fusion)
            o = 0
            for i, p in enumerate(P):
                P[i] = rows[o:o + p.shape[1] * p.shape[2]].view(p.shape)
                o += p.shape[1] * p.shape[2]
        p6 = bb.p6(P[2], n, h5, w5)
        h6, w6 = bb.p6.out_hw(h5, w5)
        p7 = bb.p7(p6, n, h6, w6, in_relu=True)
        return P + [p6, p7], dict(P=P, p6=p6, pooled=pooled, HW=(H, W))

    def forward(self, x4: torch.Tensor, H: int, W: int, N: int = 1, memory=None):
        """-> ([P3..P7] as [N,h,w,256] tensors, saved).  `memory = (memory_f16, proj)`: the memory read + sum fusion into P3..P5 as
        the hot path runs it (`eod_memory_gather_pool` + `eod_memory_project_fuse`; N == 1); None = image only."""
        p345, keep = self.forward_trunk(x4, H, W, N)
        P, tail = self.forward_tail(p345, H, W, memory)
        keep["tail"] = tail
        keep["pooled"] = tail["pooled"]
        keep["fpn"].update(P=tail["P"], p6=tail["p6"])
        return P, keep

    # ---- backward ------------------------------------------------------------------------------------------------------------
    def _relu_bw(self, g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(g)
        _lib.check(self.lib.eod_relu_backward(g.data_ptr(), y.data_ptr(), out.data_ptr(), g.numel(), ops._stream()), "eod_relu_backward")
        return out

    def _up_bw(self, g_fine: torch.Tensor, out: torch.Tensor, N: int, h: int, w: int):
        _lib.check(self.lib.eod_upsample2_sum_backward(g_fine.data_ptr(), out.data_ptr(), N, h, w, 256, 1, ops._stream()),
                   "eod_upsample2_sum_backward")

    def backward(self, saved: dict, dP: List[torch.Tensor], need_stem_grad: bool = True):
        """dP: dL/d(P3..P7) ([N,h,w,256] each) -> (grads {conv.name: (dw, db)}, dL/d(stem pre-activation) or None; without
        `need_stem_grad` the chain stops at the max pool and the stem gets no entry)."""
        grads, g_out = self.backward_tail(saved["tail"], dP)
        tg, g_stem = self.backward_trunk(saved, g_out, need_stem_grad)
        grads.update(tg)
        return grads, g_stem

    def backward_tail(self, tail: dict, dP: List[torch.Tensor]):
        """The tail half's backward: P7, P6, the memory branch's dW / db -> (grads, [g3, g4, g5] = dL/d(p3..p5 before the fusion))."""
        bb = self.bb
        P, p6 = tail["P"], tail["p6"]
        grads: Dict[str, tuple] = {}

        def put(conv, r):
            grads[conv.name] = (r["dw"], r["db"])

        # P7 = conv(relu(P6)), P6 = conv(P5 fused)
        x7 = torch.relu(p6)
        r = self._b(bb.p7)(x7, None, dP[4].contiguous())
        put(bb.p7, r)
        g6 = dP[3] + self._relu_bw(r["dx"], p6)
        r = self._b(bb.p6)(P[2].contiguous(), None, g6.contiguous(), dx_res=dP[2].contiguous())    # P5 also feeds P6: both gradients
        put(bb.p6, r)
        g_out = [dP[0].contiguous(), dP[1].contiguous(), r["dx"]]      # sum fusion: identity to the image branch
        if tail["pooled"] is not None:
            # the memory branch of the fusion: dW / db of the map_merge projections (timm.py:170-178)
            if self._merge_bw is None:
                if self.merge_weights is None:
                    raise ValueError("pass the map_merge_projection weights (merge_weights=) to back-propagate into the memory branch")
                self._merge_bw = ops.MemoryProjectorBackward(self.merge_weights, self.bb.device)
            H, W = tail["HW"]
            # the memory table is an input of the training step, not a parameter (loader.py:199-223): only dW / db are needed
            mb = self._merge_bw([t.view(-1, 256) for t in g_out], tail["pooled"], H, W, bb.map_feature_weight, need_input_grad=False)
            for i in range(3):
                grads[f"map_merge_projection{i + 1}"] = (mb["dW"][i], mb["db"][i])
        return grads, g_out

    def backward_trunk(self, saved: dict, g_out: List[torch.Tensor], need_stem_grad: bool = True, blocks: bool = True):
        """g_out: dL/d(p3..p5) of the N images of `forward_trunk` ([N,h,w,256]) -> (grads of the output / lateral convs and the
        trunk, dL/d(stem pre-activation) or None).  dW / db are sums over the N images (one launch per layer).  `blocks=False`: the
        ResNet's parameters are frozen (MODEL.FREEZE_BACKBONE): the chain stops at the laterals' weight gradients."""
        bb, N = self.bb, saved["N"]
        f = saved["fpn"]
        (c3, c4, c5), (lat3, lat4, lat5) = f["c"], f["lat"]
        (h3, w3), (h4, w4), (h5, w5), (h6, w6) = f["hw"]
        grads: Dict[str, tuple] = {}

        def put(conv, r):
            if conv.name in grads:
                raise RuntimeError(f"layer {conv.name} visited twice")
            grads[conv.name] = (r["dw"], r["db"])

        # output convs -> gradients of the merged laterals; top-down add: the coarser level also collects the 2x2 block sums
        r5 = self._b(bb.output[5])(lat5, None, g_out[2])
        put(bb.output[5], r5)
        r4 = self._b(bb.output[4])(lat4, None, g_out[1])
        put(bb.output[4], r4)
        r3 = self._b(bb.output[3])(lat3, None, g_out[0])
        put(bb.output[3], r3)
        g_lat3 = r3["dx"]
        g_lat4 = r4["dx"]
        self._up_bw(g_lat3, g_lat4, N, h4, w4)
        g_lat5 = r5["dx"]
        self._up_bw(g_lat4, g_lat5, N, h5, w5)
        gc = {}
        for l, cx, gl in ((3, c3, g_lat3), (4, c4, g_lat4), (5, c5, g_lat5)):
            # c5 feeds its lateral only: the ReLU it came out of is crossed on the way out of that layer's input-gradient launch
            r = self._b(bb.lateral[l])(cx, None, gl, dx_gate=cx if l == 5 else None, need_dx=blocks)
            put(bb.lateral[l], r)
            gc[l] = r["dx"]
        if not blocks:
            return grads, None
        # trunk, last block first; the exposed stage outputs ('layer3', 'layer4', 'layer5') collect their lateral's gradient
        blocks = bb.bottom_up.blocks
        kept = saved["blocks"]
        # gp: gradient of a block's (conv3 + shortcut), i.e. behind its final ReLU.  Inside a block every ReLU's backward and the
        # shortcut's add ride on the epilogue of the input-gradient convolution that produces the gradient (`dx_gate`, `dx_res`):
        # three launches per layer pair instead of five, and conv1's launch hands the block below its gated gradient directly.
        gp = gc[5]
        for bi in range(len(blocks) - 1, -1, -1):
            li, c1, c2, c3b, ds = blocks[bi]
            x_in, o1, o2, y, h, w, h2, w2 = kept[bi]
            r = self._b(c3b)(o2, None, gp, dx_gate=o2)        # gradient of conv2's pre-activation
            put(c3b, r)
            r2 = self._b(c2)(o1, None, r["dx"], dx_gate=o1)   # ... of conv1's
            put(c2, r2)
            shortcut = gp
            if ds is not None:
                rd = self._b(ds)(x_in, None, gp)
                put(ds, rd)
                shortcut = rd["dx"]
            # first block of stage li: its input is stage li-1's output, which the FPN reads as 'layer{li}' (timm.py:379,404) and
            # whose gradient therefore has a third term before the ReLU below it is crossed
            boundary = bi > 0 and blocks[bi - 1][0] != li and li in gc
            fuse_gate = bi > 0 and not boundary               # x_in is the block below's ReLU output
            r1 = self._b(c1)(x_in, None, r2["dx"], dx_res=shortcut, dx_gate=x_in if fuse_gate else None)
            put(c1, r1)
            g = r1["dx"]
            if boundary:
                g = g + gc[li]
                gp = self._relu_bw(g, x_in)
            else:
                gp = g
        g_stem = None
        if need_stem_grad:
            stem_out, hs, ws, pooled, hp, wp = saved["stem"]
            dpre = torch.empty_like(stem_out)
            _lib.check(self.lib.eod_maxpool3x3s2_backward(stem_out.data_ptr(), pooled.data_ptr(), g.contiguous().data_ptr(), dpre.data_ptr(),
                                                          N, hs, ws, 64, hp, wp, ops._stream()), "eod_maxpool3x3s2_backward")
            g_stem = self._relu_bw(dpre, stem_out)
            x4 = saved["x4"]
            put(bb.bottom_up.stem, self._b(bb.bottom_up.stem)(x4.view(N, -1, x4.shape[-2], 4), None, g_stem, need_dx=False))
        return grads, g_stem
