"""Backbone of the recurrent detector on the HIP kernels: timm ResNet-50 trunk + FPN + spatial-memory fusion + P6/P7.

Mirrors `build_p67_timm_fpn_backbone_recurrent` (`Detic/detic/modeling/backbone/timm.py:507-531`):
`MapTIMM/CustomResNetMap.forward` (`timm.py:277-299`), `CustomRecurrentFPN.forward` (`timm.py:91-213`) and
`LastLevelP6P7_P5` (`timm.py:347-364`).  Activations are NHWC fp32 device buffers; FrozenBatchNorm is folded
into the convs at build time; the FPN top-down add, the memory projection scale + sum fusion and all ReLUs are
epilogues of the implicit-GEMM kernel.  Outputs p3..p7 are written back-to-back into ONE [P_total, 256] buffer so
the shared-weight CenterNet head can treat the pyramid as one row list.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from .. import ops
from ..registry import BACKBONE_REGISTRY

RESNET50_LAYERS = (3, 4, 6, 3)


class ResNet50Trunk:
    def __init__(self, sd: Dict[str, torch.Tensor], device):
        base = "backbone.bottom_up.base"

        def bn(p):
            return sd[f"{p}.weight"], sd[f"{p}.bias"], sd[f"{p}.running_mean"], sd[f"{p}.running_var"]

        w, b = ops.fold_bn(sd[f"{base}.conv1.weight"], *bn(f"{base}.bn1"))
        self.stem = ops.Conv(w, b, stride=2, pad=3, device=device, cin_pad=4, name="stem")
        self.blocks: List[Tuple] = []
        for li, nblk in enumerate(RESNET50_LAYERS, start=1):
            for bi in range(nblk):
                p = f"{base}.layer{li}.{bi}"
                stride = 2 if (bi == 0 and li > 1) else 1
                w1, b1 = ops.fold_bn(sd[f"{p}.conv1.weight"], *bn(f"{p}.bn1"))
                w2, b2 = ops.fold_bn(sd[f"{p}.conv2.weight"], *bn(f"{p}.bn2"))
                w3, b3 = ops.fold_bn(sd[f"{p}.conv3.weight"], *bn(f"{p}.bn3"))
                c1 = ops.Conv(w1, b1, device=device, name=f"{p}.conv1")
                c2 = ops.Conv(w2, b2, stride=stride, pad=1, device=device, name=f"{p}.conv2")   # stride on the 3x3 (timm)
                c3 = ops.Conv(w3, b3, device=device, name=f"{p}.conv3")
                ds = None
                if f"{p}.downsample.0.weight" in sd:
                    wd, bd = ops.fold_bn(sd[f"{p}.downsample.0.weight"], *bn(f"{p}.downsample.1"))
                    ds = ops.Conv(wd, bd, stride=stride, device=device, name=f"{p}.downsample")
                self.blocks.append((li, c1, c2, c3, ds))

    def forward(self, x4: torch.Tensor, H: int, W: int, N: int = 1, plan_like_single: bool = True, keep: Optional[dict] = None):
        """x4: [N,H,W,4] normalised image(s) -> {'layer3','layer4','layer5'}: (tensor, h, w).  N > 1: every layer is ONE launch
        over the batch, planned like a single image (`plan_rows`), so each image's result is bitwise that of an N = 1 call
        (`plan_like_single = False`: planned for the rows the launch really has).  `keep`: a dict that receives every activation
        the backward needs (`modeling/backward.py`); the launches are the same."""
        def pr(conv, hh, ww):
            oh, ow = conv.out_hw(hh, ww)
            return oh * ow if (N > 1 and plan_like_single) else 0

        x = self.stem(x4, N, H, W, relu=True, plan_rows=pr(self.stem, H, W))
        h, w = self.stem.out_hw(H, W)
        stem_out, hs, ws = x, h, w
        x, h, w = ops.maxpool3x3s2(x, N, h, w, 64)
        if keep is not None:
            keep["stem"] = (stem_out, hs, ws, x, h, w)
            keep["blocks"] = []
        feats = {}
        cur_layer = 1
        for (li, c1, c2, c3, ds) in self.blocks:
            if li != cur_layer:
                feats[f"layer{cur_layer + 1}"] = (x, h, w)
                cur_layer = li
            sc = x
            if ds is not None:
                sc = ds(x, N, h, w, plan_rows=pr(ds, h, w))
            o = c1(x, N, h, w, relu=True, plan_rows=pr(c1, h, w))
            o1 = o
            o = c2(o, N, h, w, relu=True, plan_rows=pr(c2, h, w))
            h2, w2 = c2.out_hw(h, w)
            x_in = x
            x = c3(o, N, h2, w2, res=sc, res_mode=1, relu=True, plan_rows=pr(c3, h2, w2))
            if keep is not None:
                keep["blocks"].append((x_in, o1, o, x, h, w, h2, w2))
            h, w = h2, w2
        feats[f"layer{cur_layer + 1}"] = (x, h, w)
        return feats


class CustomRecurrentFPN:
    size_divisibility = 32

    def __init__(self, cfg, sd: Dict[str, torch.Tensor], device):
        self.device = device
        self.memory_type = cfg.MODEL.MEMORY_TYPE
        self.feat_fusion = cfg.MODEL.MAP_FEAT_FUSION
        self.map_feature_weight = float(cfg.MODEL.MAP_FEATURE_WEIGHT)
        if self.memory_type == "implicit_memory" and self.feat_fusion not in ("sum", "mem_only", "image_only"):
            raise ValueError(f"MODEL.MAP_FEAT_FUSION={self.feat_fusion!r} not supported (timm.py:181-186)")
        self.bottom_up = ResNet50Trunk(sd, device)
        self.lateral, self.output = {}, {}
        for l in (3, 4, 5):
            self.lateral[l] = ops.Conv(sd[f"backbone.fpn_lateral{l}.weight"], sd[f"backbone.fpn_lateral{l}.bias"], device=device,
                                       name=f"fpn_lateral{l}")
            self.output[l] = ops.Conv(sd[f"backbone.fpn_output{l}.weight"], sd[f"backbone.fpn_output{l}.bias"], pad=1, device=device,
                                      name=f"fpn_output{l}")
        self.p6 = ops.Conv(sd["backbone.top_block.p6.weight"], sd["backbone.top_block.p6.bias"], stride=2, pad=1, device=device, name="p6")
        self.p7 = ops.Conv(sd["backbone.top_block.p7.weight"], sd["backbone.top_block.p7.bias"], stride=2, pad=1, device=device, name="p7")
        self.merge = ops.MemoryProjector([sd[f"backbone.map_merge_projection{i}.weight"] for i in (1, 2, 3)],
                                         [sd[f"backbone.map_merge_projection{i}.bias"] for i in (1, 2, 3)], device)
        self._plans = {}
        # True (default): 4x4 pooling blocks are summed pixel by pixel in F.avg_pool2d's order -- bit-identical to the reference's
        # rounding (timm.py:147-168) on every input.  False: per distinct cell (count x row), ~9 us per frame faster at 640x640 but
        # identical only when a window's exponents span <= 9 bits (tests/test_kernels_gpu.py asserts > 0.9995 of the values).
        self.pool_in_torch_order = True

    def level_shapes(self, H: int, W: int) -> List[Tuple[int, int]]:
        hw = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
        h6, w6 = self.p6.out_hw(*hw[2])
        h7, w7 = self.p7.out_hw(h6, w6)
        return hw + [(h6, w6), (h7, w7)]

    def _plan(self, H: int, W: int, which: int = 0):
        """Buffers of one pyramid; `which` in {0, 1}: two sets, so that the next frame's top-down pass can be written while the
        current frame's ROI passes still read theirs."""
        key = (H, W, which)
        if key not in self._plans:
            shapes = self.level_shapes(H, W)
            off = [0]
            for (h, w) in shapes:
                off.append(off[-1] + h * w)
            feats = torch.empty((off[-1], 256), dtype=torch.float32, device=self.device)
            views = [feats[off[i]:off[i + 1]].view(1, shapes[i][0], shapes[i][1], 256) for i in range(5)]
            pooled = torch.empty((ops.pooled_rows(H, W), 512), dtype=torch.float16, device=self.device)
            self._plans[key] = (shapes, off, feats, views, pooled)
        return self._plans[key]

    def top_down(self, c: dict, H: int, W: int, which: int = 0):
        """Memory-independent half (timm.py:118-136): lateral 1x1, + nearest x2 of the coarser level, 3x3 output -> P3..P5 of
        buffer set `which`."""
        shapes, off, feats, views, pooled = self._plan(H, W, which)
        (c5, h5, w5), (c4, h4, w4), (c3, h3, w3) = c["layer5"], c["layer4"], c["layer3"]
        assert (h3, w3) == shapes[0] and (h5, w5) == shapes[2]
        lat5 = self.lateral[5](c5, 1, h5, w5)
        self.output[5](lat5, 1, h5, w5, out=views[2])
        lat4 = self.lateral[4](c4, 1, h4, w4, res=lat5, res_mode=2)
        self.output[4](lat4, 1, h4, w4, out=views[1])
        lat3 = self.lateral[3](c3, 1, h3, w3, res=lat4, res_mode=2)
        self.output[3](lat3, 1, h3, w3, out=views[0])

    def top_down_batched(self, c: dict, H: int, W: int, N: int):
        """`top_down` for N images in one launch per layer -> level-major tensors (P3 [N,h3,w3,256], P4, P5); planned like one
        image.  The caller copies each image's slices into that scene's own pyramid set (modeling/batched.py)."""
        (c5, h5, w5), (c4, h4, w4), (c3, h3, w3) = c["layer5"], c["layer4"], c["layer3"]
        lat5 = self.lateral[5](c5, N, h5, w5, plan_rows=h5 * w5)
        p5 = self.output[5](lat5, N, h5, w5, plan_rows=h5 * w5)
        lat4 = self.lateral[4](c4, N, h4, w4, res=lat5, res_mode=2, plan_rows=h4 * w4)
        p4 = self.output[4](lat4, N, h4, w4, plan_rows=h4 * w4)
        lat3 = self.lateral[3](c3, N, h3, w3, res=lat4, res_mode=2, plan_rows=h3 * w3)
        p3 = self.output[3](lat3, N, h3, w3, plan_rows=h3 * w3)
        return [p3, p4, p5]

    def fuse_memory_and_top(self, H: int, W: int, memory_f16: Optional[torch.Tensor], proj: Optional[torch.Tensor], which: int = 0,
                            err: Optional[torch.Tensor] = None):
        """Memory read + fusion into P3..P5 (timm.py:142-192), then P6/P7 on the fused P5 (timm.py:200-205, 359-364)."""
        shapes, off, feats, views, pooled = self._plan(H, W, which)
        h5, w5 = shapes[2]
        if self.memory_type == "implicit_memory" and self.feat_fusion != "image_only":
            if memory_f16 is None or proj is None:
                raise ValueError("implicit_memory needs the fp16 memory table and proj_indices")
            # P3..P5 are the first rows of the pyramid's row list, in the order the pooled rows are written
            ops.memory_gather_pool(memory_f16, proj, H, W, out=pooled, err=err, torch_order=self.pool_in_torch_order)
            self.merge(pooled, feats, H, W, self.map_feature_weight, self.feat_fusion)
        self.p6(views[2], 1, h5, w5, out=views[3])
        self.p7(views[3], 1, shapes[3][0], shapes[3][1], in_relu=True, out=views[4])
        return feats, views, shapes, off

    def forward(self, x4: torch.Tensor, H: int, W: int, memory_f16: Optional[torch.Tensor], proj: Optional[torch.Tensor],
                which: int = 0, err: Optional[torch.Tensor] = None):
        """-> (feats [P_total,256], level views, level shapes, level offsets)."""
        self.top_down(self.bottom_up.forward(x4, H, W), H, W, which)
        return self.fuse_memory_and_top(H, W, memory_f16, proj, which, err)


@BACKBONE_REGISTRY.register()
def build_p67_timm_fpn_backbone_recurrent(cfg, sd, device):
    if cfg.MODEL.TIMM.BASE_NAME != "resnet50_in21k_map":
        # the reference crashes on an undefined `new_memory` for any other base (custom_rcnn.py:560-567,580)
        raise ValueError("MODEL.TIMM.BASE_NAME must be 'resnet50_in21k_map' for the recurrent path")
    return CustomRecurrentFPN(cfg, sd, device)
