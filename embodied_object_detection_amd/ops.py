"""Thin host wrappers over the C ABI: torch supplies device memory and the stream, nothing else.

Every function enqueues HIP kernels on torch's current stream and returns torch tensors that alias the
buffers the kernels write.  No arithmetic is done with torch ops here.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import EodConvDesc, EodDetDesc, EodMemWriteDesc, EodProposalDesc, check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Raw handle of torch's current stream on the current device.  The private fast path avoids building a Stream object per
    launch (torch.cuda.current_stream() was 45 % of the host time of a frame: ~180 launches)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.EodError("HIP product path needs device tensors (no CPU fallback)")


class Workspace:
    """Grow-only scratch buffer per device (split-K slabs, selection scratch)."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        # one buffer per (device, stream): kernels of different streams may run concurrently
        key = (device.index if isinstance(device, torch.device) else str(device), _stream())
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


_conv_ws = Workspace()


# ----------------------------------------------------------------------------------------------------
# weight preparation (host, once per model build)
# ----------------------------------------------------------------------------------------------------
_CONV_MATH = {"fp32": 0, "bf16x3": 1}


def set_conv_math(mode: str) -> str:
    """Arithmetic of every conv / linear launch from now on (process-wide): "fp32" (fp32 MFMA, default) or "bf16x3" (three-way
    bf16 operand split on the bf16 MFMA pipe, fp32 accumulate; include/eod_hip.h eod_set_conv_math).  Returns the previous mode."""
    if mode not in _CONV_MATH:
        raise ValueError(f"conv math must be one of {sorted(_CONV_MATH)}, got {mode!r}")
    prev = _lib.load().eod_set_conv_math(_CONV_MATH[mode])
    check(min(prev, 0), "eod_set_conv_math")
    return {v: k for k, v in _CONV_MATH.items()}[prev]


def get_conv_math() -> str:
    return {v: k for k, v in _CONV_MATH.items()}[_lib.load().eod_get_conv_math()]


def fold_bn(w: torch.Tensor, bn_w, bn_b, bn_mean, bn_var, eps: float = 1e-5):
    """FrozenBatchNorm2d folded into the preceding bias-free conv."""
    scale = bn_w / torch.sqrt(bn_var + eps)
    return w * scale.view(-1, 1, 1, 1), bn_b - bn_mean * scale


def pack_conv_weight(w_oihw: torch.Tensor, cin_pad: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """OIHW -> [Cout, Kpad] with k = (ky, kx, c), c fastest; K padded with zeros to a multiple of 32."""
    O, I, KH, KW = w_oihw.shape
    w = w_oihw.permute(0, 2, 3, 1).contiguous()  # O,KH,KW,I
    if cin_pad is not None and cin_pad != I:
        wp = torch.zeros((O, KH, KW, cin_pad), dtype=w.dtype)
        wp[..., :I] = w
        w = wp
    K = w.shape[1] * w.shape[2] * w.shape[3]
    Kpad = (K + 31) // 32 * 32
    out = torch.zeros((O, Kpad), dtype=torch.float32)
    out[:, :K] = w.reshape(O, K)
    return out.contiguous(), Kpad


class Conv:
    """One prepared conv / linear layer living on the device."""

    def __init__(self, w_oihw: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, pad: int = 0,
                 device="cuda", cin_pad: Optional[int] = None, deconv: bool = False, name: str = ""):
        self.name = name
        if deconv:
            # ConvTranspose2d(k=2, s=2) weight [Cin, Cout, 2, 2] -> rows n = (dy*2+dx)*Cout + co, K = Cin
            Cin, Cout, kh, kw = w_oihw.shape
            assert kh == 2 and kw == 2
            w = w_oihw.permute(2, 3, 1, 0).reshape(4 * Cout, Cin, 1, 1).contiguous()
            self.out_mode = 1
            self.KH = self.KW = 1
            self.stride, self.pad = 1, 0
            self.Cin, self.Cout = Cin, 4 * Cout
        else:
            w = w_oihw
            self.out_mode = 0
            self.Cout, cin, self.KH, self.KW = w.shape
            self.Cin = cin_pad or cin
            self.stride, self.pad = stride, pad
        packed, self.Kpad = pack_conv_weight(w, cin_pad)
        self.tap4 = 1 if self.Cin == 4 else 0
        if not self.tap4 and self.Cin % 32 != 0:
            raise ValueError(f"{name}: Cin={self.Cin} must be a multiple of 32 (or 4 for the stem)")
        self.w = packed.to(device)
        self.bias = None if bias is None else bias.detach().to(torch.float32).contiguous().to(device)
        self.desc = EodConvDesc()
        self._lib = _lib.load()
        self.w_split = None     # bf16x3 pieces of the weights, made on first use in that arithmetic mode
        self.event_log = None   # bench.py: list that receives (start_event, end_event, event_tag) per launch
        self.event_tag = None   # what the caller wants to know the launch by (no device work: a clone of m_count would be a launch)
        self.lds_reserve = 0    # EodConvDesc.lds_reserve: caps this layer's workgroups per CU (see include/eod_hip.h)
        self.prefetch2 = 0      # EodConvDesc.prefetch2: two chunks of operands in flight (small launches)

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        return ((H + 2 * self.pad - self.KH) // self.stride + 1, (W + 2 * self.pad - self.KW) // self.stride + 1)

    def __call__(self, x: torch.Tensor, N: int, H: int, W: int, *, res: Optional[torch.Tensor] = None, res_mode: int = 0,
                 relu: bool = False, in_relu: bool = False, out_scale: float = 1.0, m_count: Optional[torch.Tensor] = None,
                 m_unit: int = 0, out: Optional[torch.Tensor] = None, force_tile: int = 0, force_splitk: int = 0,
                 levels: Optional[Tuple[Sequence[int], Sequence[Tuple[int, int]]]] = None,
                 fuse: Optional[Tuple[torch.Tensor, float, Optional[torch.Tensor]]] = None, presplit: bool = True,
                 plan_rows: int = 0, gn_stats: Optional[torch.Tensor] = None, gn_groups: int = 32,
                 split: Optional[Tuple[int, torch.Tensor]] = None, m_segments: int = 0,
                 gate: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`levels=(row_offsets, [(h, w), ...])` runs the layer once over a whole feature pyramid stored as one row list.
        `split=(n0, out2)`: the layer is two stacked linear layers; columns [0, n0) go to `out` [rows, n0] without the ReLU, columns
        [n0, Cout) to `out2` [rows, Cout - n0] with it (EodConvDesc.split_n).
        `m_segments` = B: the N images are B unit lists back to back, `m_count` holds B counts (EodConvDesc.m_segments).
        `gn_stats` (pyramid mode; the workspace of the `groupnorm_relu` call that follows): when the layer's plan reduces split-K
        slabs, that reduce also writes GroupNorm's partial sums into it and `self.gn_fused` is set (pass it as `partial_ready`).
        `gate` [N,OH,OW,Cout]: the output is zeroed where gate <= 0 (EodConvDesc.gate: a ReLU's backward on the way out).
        `fuse=(pred_w [Cout/4], pred_b, out_units or None)` (deconv layers only): ConvTranspose + ReLU + 1x1 predictor + sigmoid
        in one launch, `out` = [units, 2H, 2W] probabilities (out_mode 2 of include/eod_hip.h)."""
        _need_cuda(x, res, out, gate)
        if fuse is not None:
            if self.out_mode != 1 or out is None:
                raise ValueError("fuse= needs a deconv layer and an explicit probability buffer `out`")
            _need_cuda(fuse[0], fuse[2])
        OH, OW = self.out_hw(H, W) if levels is None else (0, 0)
        if levels is not None and out is None:
            out = torch.empty((levels[0][-1], self.Cout), dtype=torch.float32, device=x.device)
        if out is None:
            if self.out_mode == 1:
                out = torch.empty((N, 2 * OH, 2 * OW, self.Cout // 4), dtype=torch.float32, device=x.device)
            else:
                out = torch.empty((N, OH, OW, self.Cout), dtype=torch.float32, device=x.device)
        d = self.desc
        d.x, d.w, d.bias, d.res, d.y = x.data_ptr(), self.w.data_ptr(), _ptr(self.bias), _ptr(res), out.data_ptr()
        d.m_count, d.m_unit, d.m_segments = _ptr(m_count), m_unit, int(m_segments)
        d.gate = _ptr(gate)
        d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout = N, H, W, self.Cin, OH, OW, self.Cout
        d.KH, d.KW, d.stride, d.pad, d.Kpad = self.KH, self.KW, self.stride, self.pad, self.Kpad
        d.relu, d.res_mode, d.in_relu, d.out_mode, d.tap4 = int(relu), res_mode, int(in_relu), self.out_mode, self.tap4
        if fuse is not None:
            d.out_mode, d.fuse_w, d.fuse_b, d.out_units = 2, fuse[0].data_ptr(), float(fuse[1]), _ptr(fuse[2])
        else:
            d.fuse_w, d.fuse_b, d.out_units = None, 0.0, None
        d.force_tile, d.force_splitk, d.out_scale = force_tile, force_splitk, out_scale
        d.plan_rows = plan_rows
        d.lds_reserve = self.lds_reserve
        d.prefetch2 = self.prefetch2
        d.gn_partial, d.gn_groups = None, 0
        self.gn_fused = False
        if split is not None:
            _need_cuda(split[1])
            d.split_n, d.y2 = int(split[0]), split[1].data_ptr()
        else:
            d.split_n, d.y2 = 0, None
        if levels is not None:
            off, shapes = levels
            d.levels = len(shapes)
            for i, o in enumerate(off):
                d.level_off[i] = o
            for i, (h, w) in enumerate(shapes):
                d.level_h[i], d.level_w[i] = h, w
        else:
            d.levels = 0
        use_split = force_tile // 10 == 5 if force_tile else (get_conv_math() == "bf16x3")
        if use_split and self.w_split is None and not self.tap4:
            nb = self._lib.eod_conv_split_weights_bytes(self.Cout, self.Kpad)
            self.w_split = torch.empty((nb,), dtype=torch.uint8, device=self.w.device)
            check(self._lib.eod_conv_split_weights_bf16x3(self.w.data_ptr(), self.Cout, self.Kpad, self.w_split.data_ptr(), _stream()),
                  f"eod_conv_split_weights_bf16x3[{self.name}]")
        d.w_split = self.w_split.data_ptr() if (use_split and self.w_split is not None and presplit) else None
        if gn_stats is not None and levels is not None and self._lib.eod_conv2d_gn_fused(C.byref(d)):
            d.gn_partial = gn_stats.data_ptr() + self._lib.eod_groupnorm_partial_offset(len(levels[1]), gn_groups)
            d.gn_groups = gn_groups
            self.gn_fused = True
        d.workspace, d.workspace_bytes = None, 0
        need = self._lib.eod_conv2d_workspace_bytes(C.byref(d))
        if need:
            ws = _conv_ws.get(need, x.device)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        if self.event_log is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(self._lib.eod_conv2d(C.byref(d), _stream()), f"eod_conv2d[{self.name}]")
            e1.record()
            self.event_log.append((e0, e1, self.event_tag))
            return out
        check(self._lib.eod_conv2d(C.byref(d), _stream()), f"eod_conv2d[{self.name}]")
        return out


# ----------------------------------------------------------------------------------------------------
# elementwise / pooling
# ----------------------------------------------------------------------------------------------------
def preprocess_image(img_u8_chw: torch.Tensor, mean: Sequence[float], std: Sequence[float], div: int = 32,
                     out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, int, int]:
    """`out` (optional): a [1,Hp,Wp,4] slice of a batch buffer to write into."""
    _need_cuda(img_u8_chw)
    assert img_u8_chw.dtype == torch.uint8 and img_u8_chw.dim() == 3 and img_u8_chw.is_contiguous()
    _, H, W = img_u8_chw.shape
    Hp, Wp = (H + div - 1) // div * div, (W + div - 1) // div * div
    if out is None:
        out = torch.empty((1, Hp, Wp, 4), dtype=torch.float32, device=img_u8_chw.device)
    assert out.numel() == Hp * Wp * 4 and out.is_contiguous()
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    check(_lib.load().eod_preprocess_image(img_u8_chw.data_ptr(), out.data_ptr(), H, W, Hp, Wp, m, s, _stream()),
          "eod_preprocess_image")
    return out, Hp, Wp


def maxpool3x3s2(x: torch.Tensor, N: int, H: int, W: int, Cc: int) -> Tuple[torch.Tensor, int, int]:
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, OH, OW, Cc), dtype=torch.float32, device=x.device)
    check(_lib.load().eod_maxpool3x3s2(x.data_ptr(), y.data_ptr(), N, H, W, Cc, OH, OW, _stream()), "eod_maxpool3x3s2")
    return y, OH, OW


def groupnorm_workspace(level_off: Sequence[int], device, groups: int = 32) -> torch.Tensor:
    lo = (C.c_int32 * len(level_off))(*level_off)
    n = _lib.load().eod_groupnorm_workspace_bytes(lo, len(level_off) - 1, groups)
    return torch.empty(((n + 7) // 8,), dtype=torch.float64, device=device)


def groupnorm_relu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, level_off: Sequence[int], Cc: int,
                   stats: torch.Tensor, groups: int = 32, eps: float = 1e-5, out: Optional[torch.Tensor] = None,
                   partial_ready: bool = False) -> torch.Tensor:
    """`partial_ready`: the conv that produced `x` already wrote the partial sums into `stats` (Conv(..., gn_stats=stats))."""
    if out is None:
        out = torch.empty_like(x)
    lo = (C.c_int32 * len(level_off))(*level_off)
    check(_lib.load().eod_groupnorm_relu(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), lo, len(level_off) - 1,
                                         Cc, groups, eps, stats.data_ptr(), int(partial_ready), _stream()), "eod_groupnorm_relu")
    return out


def groupnorm_relu_backward(x: torch.Tensor, y: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, level_off: Sequence[int], Cc: int,
                            fwd_stats: torch.Tensor, groups: int = 32, eps: float = 1e-5):
    """Backward of `groupnorm_relu` (its input `x`, output `y`, the `stats` workspace of that call) -> (dx, dgamma, dbeta)."""
    _need_cuda(x, y, dy, gamma, fwd_stats)
    lo = (C.c_int32 * len(level_off))(*level_off)
    lib = _lib.load()
    ws = torch.empty((lib.eod_groupnorm_backward_workspace_bytes(lo, len(level_off) - 1, Cc) // 8,), dtype=torch.float64, device=x.device)
    dx = torch.empty_like(x)
    dgamma = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    dbeta = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    check(lib.eod_groupnorm_relu_backward(x.data_ptr(), y.data_ptr(), dy.data_ptr(), gamma.data_ptr(), lo, len(level_off) - 1, Cc, groups, eps,
                                          fwd_stats.data_ptr(), ws.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _stream()),
          "eod_groupnorm_relu_backward")
    return dx, dgamma, dbeta


def mask_predictor_sigmoid(x: torch.Tensor, w: torch.Tensor, bias: float, rows: int, Cc: int, count: Optional[torch.Tensor],
                           unit_rows: int, out: Optional[torch.Tensor] = None, out_units: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(_lib.load().eod_mask_predictor_sigmoid(x.data_ptr(), w.data_ptr(), bias, out.data_ptr(), rows, Cc, _ptr(count),
                                                 unit_rows, _ptr(out_units), _stream()), "eod_mask_predictor_sigmoid")
    return out


def roi_align(p3, p4, p5, h3: int, w3: int, Cc: int, boxes: torch.Tensor, count: Optional[torch.Tensor], R_cap: int, S: int,
              out: Optional[torch.Tensor] = None, box_rows: Optional[torch.Tensor] = None, batch: int = 1,
              boxes_per_image: int = 0, refine=None) -> torch.Tensor:
    """`batch` > 1: p3..p5 are [batch,h,w,C]; box j is pooled from image j // boxes_per_image (include/eod_hip.h).
    `refine = (deltas, ld, weights, clip, img_w, img_h, boxes_out)`: the ROI boxes are apply_deltas(boxes, deltas), also written to
    `boxes_out` (EodBoxRefine: the cascade's next-stage proposals without their own launch)."""
    if out is None:
        out = torch.empty((R_cap, S, S, Cc), dtype=torch.float32, device=p3.device)
    ref = None
    if refine is not None:
        deltas, ld, (wx, wy, ww, wh), clip, img_w, img_h, boxes_out = refine
        _need_cuda(deltas, boxes_out)
        ref = _lib.EodBoxRefine(deltas.data_ptr(), ld, wx, wy, ww, wh, int(clip), img_w, img_h, boxes_out.data_ptr())
    check(_lib.load().eod_roi_align(p3.data_ptr(), p4.data_ptr(), p5.data_ptr(), h3, w3, Cc, boxes.data_ptr(), _ptr(box_rows),
                                    _ptr(count), R_cap, S, out.data_ptr(), batch, boxes_per_image, None if ref is None else C.byref(ref),
                                    _stream()), "eod_roi_align")
    return out


def roi_align_backward(dp3, dp4, dp5, h3: int, w3: int, Cc: int, boxes: torch.Tensor, count: Optional[torch.Tensor], R_cap: int, S: int,
                       g: torch.Tensor):
    """Adds the pooling's gradient into the pyramid levels' gradients dp3..dp5 ([h,w,C] each, one image); g [R,S,S,C]."""
    _need_cuda(dp3, dp4, dp5, boxes, g)
    assert g.is_contiguous() and tuple(g.shape) == (R_cap, S, S, Cc)
    check(_lib.load().eod_roi_align_backward(dp3.data_ptr(), dp4.data_ptr(), dp5.data_ptr(), h3, w3, Cc, boxes.data_ptr(), _ptr(count),
                                             R_cap, S, g.data_ptr(), _stream()), "eod_roi_align_backward")


def concat_lists(lists: torch.Tensor, counts: torch.Tensor, cap_in: int, id_stride: int, batch: int, out: torch.Tensor,
                 out_count: torch.Tensor):
    """The scene-local index lists of a batch as one list of global indices b * id_stride + lists[b][k] (eod_concat_lists)."""
    check(_lib.load().eod_concat_lists(lists.data_ptr(), counts.data_ptr(), cap_in, id_stride, batch, out.data_ptr(), out_count.data_ptr(),
                                       _stream()), "eod_concat_lists")
    return out, out_count


def unique_rows(rows: torch.Tensor, count: torch.Tensor, K_cap: int, R_cap: int, out_rows: torch.Tensor, out_count: torch.Tensor):
    check(_lib.load().eod_unique_rows(rows.data_ptr(), count.data_ptr(), K_cap, R_cap, out_rows.data_ptr(), out_count.data_ptr(),
                                      _stream()), "eod_unique_rows")


# ----------------------------------------------------------------------------------------------------
# selection
# ----------------------------------------------------------------------------------------------------
class ProposalDecoder:
    """CenterNet.inference on device (centernet.py:603-745)."""

    def __init__(self, level_hw: Sequence[Tuple[int, int]], strides: Sequence[int], scales: Sequence[float], score_thresh: float,
                 pre_nms_topk: int, post_nms_topk: int, nms_thresh: float, cap: int, device, head_stride: int = 5, batch: int = 1):
        """`batch` > 1: B scenes in lock-step; `head_out` is level major over the scenes, the outputs are [B*cap,4] / [B*cap] / [B]."""
        self.lib = _lib.load()
        d = EodProposalDesc()
        off = [0]
        for (h, w) in level_hw:
            off.append(off[-1] + h * w)
        d.levels = len(level_hw)
        for i, o in enumerate(off):
            d.level_off[i] = o
        for i, (h, w) in enumerate(level_hw):
            d.level_w[i] = w
            d.level_stride[i] = strides[i]
            d.level_scale[i] = scales[i]
        d.head_stride = head_stride
        d.score_thresh, d.pre_nms_topk, d.post_nms_topk, d.nms_thresh, d.cap = score_thresh, pre_nms_topk, post_nms_topk, nms_thresh, cap
        nbytes = self.lib.eod_proposals_workspace_bytes(off[-1], d.levels, pre_nms_topk, batch)
        d.batch = batch
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        self.boxes = torch.zeros((batch * cap, 4), dtype=torch.float32, device=device)
        self.scores = torch.zeros((batch * cap,), dtype=torch.float32, device=device)
        self.count = torch.zeros((batch,), dtype=torch.int32, device=device)
        d.out_boxes, d.out_scores, d.out_count = self.boxes.data_ptr(), self.scores.data_ptr(), self.count.data_ptr()
        d.workspace, d.workspace_bytes = self.ws.data_ptr(), nbytes
        self.desc = d
        self.total = off[-1]
        self.level_off = off

    def __call__(self, head_out: torch.Tensor):
        self.desc.head_out = head_out.data_ptr()
        check(self.lib.eod_centernet_proposals(C.byref(self.desc), _stream()), "eod_centernet_proposals")
        return self.boxes, self.scores, self.count


class DetectionSelector:
    """detectron2 fast_rcnn_inference (single image) on device: ONE launch.  `unique=True`: the launch also writes torch.unique of
    the kept proposal rows (`uniq_rows`, `uniq_count`; custom_rcnn.py:875).  `groups=True`: it also groups the detections by
    proposal row -- detections of one row carry the same class-agnostic box: `rep_of[k]` = the first detection of k's group,
    `rep_list` / `rep_count` = the group representatives."""

    def __init__(self, R_cap: int, C1: int, topk: int, device, unique: bool = False, groups: bool = False, batch: int = 1):
        """`batch` > 1: B scenes, one workgroup each; every buffer is B single-scene buffers back to back, indices stay scene-local."""
        self.lib = _lib.load()
        self.R_cap, self.C1, self.topk, self.batch = R_cap, C1, topk, batch
        nbytes = self.lib.eod_detections_workspace_bytes(R_cap, C1)
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        B = batch
        self.boxes = torch.zeros((B * topk, 4), dtype=torch.float32, device=device)
        self.scores = torch.zeros((B * topk,), dtype=torch.float32, device=device)
        self.classes = torch.zeros((B * topk,), dtype=torch.int32, device=device)
        self.rows = torch.zeros((B * topk,), dtype=torch.int32, device=device)
        self.count = torch.zeros((B,), dtype=torch.int32, device=device)
        d = EodDetDesc()
        d.R_cap, d.C1, d.topk, d.batch = R_cap, C1, topk, batch
        d.out_boxes, d.out_scores, d.out_classes = self.boxes.data_ptr(), self.scores.data_ptr(), self.classes.data_ptr()
        d.out_rows, d.out_count = self.rows.data_ptr(), self.count.data_ptr()
        d.workspace, d.workspace_bytes = self.ws.data_ptr(), nbytes
        self.uniq_rows = self.uniq_count = None
        if unique:
            self.uniq_rows = torch.zeros((B * R_cap,), dtype=torch.int32, device=device)
            self.uniq_count = torch.zeros((B,), dtype=torch.int32, device=device)
            d.out_unique_rows, d.out_unique_count, d.unique_cap = self.uniq_rows.data_ptr(), self.uniq_count.data_ptr(), R_cap
        self.rep_of = self.rep_list = self.rep_count = None
        if groups:
            self.rep_of = torch.zeros((B * topk,), dtype=torch.int32, device=device)
            self.rep_list = torch.zeros((B * topk,), dtype=torch.int32, device=device)
            self.rep_count = torch.zeros((B,), dtype=torch.int32, device=device)
            d.out_rep_of, d.out_rep_list, d.out_rep_count = self.rep_of.data_ptr(), self.rep_list.data_ptr(), self.rep_count.data_ptr()
        self.desc = d

    def __call__(self, boxes: torch.Tensor, scores: torch.Tensor, count: Optional[torch.Tensor], img_w: float, img_h: float,
                 score_thresh: float, nms_thresh: float):
        d = self.desc
        d.boxes, d.scores, d.count = boxes.data_ptr(), scores.data_ptr(), _ptr(count)
        d.img_w, d.img_h, d.score_thresh, d.nms_thresh = img_w, img_h, score_thresh, nms_thresh
        check(self.lib.eod_fast_rcnn_inference(C.byref(d), _stream()), "eod_fast_rcnn_inference")
        return self.boxes, self.scores, self.classes, self.rows, self.count


def zs_classify(feat, zs, prob_acc, accumulate: bool, featn_out, count, R_cap: int, C1: int, temp: float = 50.0, zs_mem=None,
                prop_scores=None, mem_scores_out=None, final_inv_stages: float = 0.0, batch: int = 1):
    """`zs_mem` + `prop_scores` + `mem_scores_out`: also the memory update's CLIP re-score (what `memory_scores` computes) in the same
    launch; `final_inv_stages` > 0 (last cascade stage): also the cascade score fusion (what `cascade_scores` does)."""
    st = _lib.load().eod_zs_classify(feat.data_ptr(), zs.data_ptr(), prob_acc.data_ptr(), int(accumulate), _ptr(featn_out), _ptr(count),
                                     R_cap, 512, C1, temp, _ptr(zs_mem), _ptr(prop_scores), _ptr(mem_scores_out),
                                     float(final_inv_stages), batch, _stream())
    if st == -5:
        raise _lib.EodError(f"eod_zs_classify: {C1 - 1} classes + background do not fit the kernel's LDS-staged class matrix (at most 23 "
                            "classes): a RESET_CLS_TESTS / TEST_NUM_CLASSES vocabulary of this size is not supported on the HIP path")
    check(st, "eod_zs_classify")


def cascade_stage_tail(feat, zs, prob_acc, accumulate: bool, featn_out, count, R_cap: int, C1: int, temp: float, hb, bb2: "Conv", boxes_in,
                       boxes_out, weights, clip: bool, img_w: float, img_h: float, zs_mem=None, prop_scores=None, mem_scores_out=None,
                       final_inv_stages: float = 0.0, deltas_out=None, batch: int = 1):
    """`zs_classify` + bbox_pred.2 + apply_deltas of one cascade stage in ONE launch (`eod_cascade_stage_tail`); `bb2` is the stage's
    1024 -> 4 layer (its packed weight rows and bias are read in place)."""
    d = _lib.EodStageTailDesc()
    d.feat, d.zs, d.prob_acc, d.accumulate, d.feat_norm_out, d.count = feat.data_ptr(), zs.data_ptr(), prob_acc.data_ptr(), int(accumulate), _ptr(featn_out), _ptr(count)
    d.R_cap, d.D, d.C1, d.temp = R_cap, 512, C1, temp
    d.zs_mem, d.prop_scores, d.mem_scores_out, d.final_inv_stages, d.batch = _ptr(zs_mem), _ptr(prop_scores), _ptr(mem_scores_out), float(final_inv_stages), batch
    d.hb, d.w2, d.b2, d.hb_dim, d.w2_ld = hb.data_ptr(), bb2.w.data_ptr(), bb2.bias.data_ptr(), bb2.Cin, bb2.Kpad
    d.boxes_in, d.boxes_out, d.deltas_out = boxes_in.data_ptr(), boxes_out.data_ptr(), _ptr(deltas_out)
    d.wx, d.wy, d.ww, d.wh = weights
    d.clip, d.img_w, d.img_h = int(clip), img_w, img_h
    st = _lib.load().eod_cascade_stage_tail(C.byref(d), _stream())
    if st == -5:
        raise _lib.EodError(f"eod_cascade_stage_tail: {C1 - 1} classes exceed the classifier kernel's capacity")
    check(st, "eod_cascade_stage_tail")


def apply_deltas(deltas, ld: int, boxes, out, count, R_cap: int, weights, clip: bool, img_w: float, img_h: float, batch: int = 1):
    wx, wy, ww, wh = weights
    check(_lib.load().eod_apply_deltas(deltas.data_ptr(), ld, boxes.data_ptr(), out.data_ptr(), _ptr(count), R_cap, wx, wy, ww, wh,
                                       int(clip), img_w, img_h, batch, _stream()), "eod_apply_deltas")


def cascade_scores(prob_acc, prop_scores, count, R_cap: int, C1: int, inv_stages: float):
    check(_lib.load().eod_cascade_scores(prob_acc.data_ptr(), prop_scores.data_ptr(), _ptr(count), R_cap, C1, inv_stages, _stream()),
          "eod_cascade_scores")


def memory_scores(featn, zs, prop_scores, scores_out, count, R_cap: int, C1: int):
    check(_lib.load().eod_memory_scores(featn.data_ptr(), zs.data_ptr(), prop_scores.data_ptr(), scores_out.data_ptr(), _ptr(count),
                                        R_cap, 512, C1, _stream()), "eod_memory_scores")


def detector_postprocess(boxes, scores, classes, count, cap: int, sx: float, sy: float, out_w: float, out_h: float, ob, os_, oc, osrc,
                         ocount, remap=None, batch: int = 1):
    """`remap` (int32 [cap], optional): osrc[q] = remap[index of the kept detection] -- the detection whose mask stands for it."""
    check(_lib.load().eod_detector_postprocess(boxes.data_ptr(), scores.data_ptr(), classes.data_ptr(), _ptr(count), cap, sx, sy,
                                               out_w, out_h, ob.data_ptr(), os_.data_ptr(), oc.data_ptr(), osrc.data_ptr(),
                                               ocount.data_ptr(), _ptr(remap), batch, _stream()), "eod_detector_postprocess")


def paste_masks(prob, boxes, rows, count, K_cap: int, H: int, W: int, thr: float, out: torch.Tensor, batch: int = 1,
                prob_units: int = 0):
    check(_lib.load().eod_paste_masks(prob.data_ptr(), boxes.data_ptr(), _ptr(rows), _ptr(count), K_cap, H, W, thr, out.data_ptr(),
                                      batch, prob_units, _stream()), "eod_paste_masks")
    return out


# ----------------------------------------------------------------------------------------------------
# spatial memory
# ----------------------------------------------------------------------------------------------------
def unproject_grid_index(depth: torch.Tensor, T: np.ndarray, intr, proj_shift, map_shift, cell: float, map_w: int, map_h: int,
                         order: int = 0, want_xyz: bool = False):
    _need_cuda(depth)
    H, W = depth.shape
    idx = torch.empty((H, W), dtype=torch.int32, device=depth.device)
    xyz = torch.empty((H, W, 3), dtype=torch.float32, device=depth.device) if want_xyz else None
    T16 = (C.c_float * 16)(*np.asarray(T, dtype=np.float32).reshape(-1).tolist())
    ps = (C.c_float * 3)(*np.asarray(proj_shift, dtype=np.float32).tolist())
    ms = (C.c_float * 3)(*np.asarray(map_shift, dtype=np.float32).tolist())
    fx, fy, cx, cy = [float(np.float32(v)) for v in intr]
    check(_lib.load().eod_unproject_grid_index(depth.data_ptr(), H, W, T16, fx, fy, cx, cy, ps, ms, float(np.float32(cell)), map_w,
                                               map_h, order, _ptr(xyz), idx.data_ptr(), _stream()), "eod_unproject_grid_index")
    return (idx, xyz) if want_xyz else idx


def memory_normalize_f16(mem: torch.Tensor, obs: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    N, D = mem.shape
    if out is None:
        out = torch.empty((N, D), dtype=torch.float16, device=mem.device)
    check(_lib.load().eod_memory_normalize_f16(mem.data_ptr(), obs.data_ptr(), out.data_ptr(), N, D, _stream()),
          "eod_memory_normalize_f16")
    return out


def memory_normalize_dirty_f16(mem: torch.Tensor, obs: torch.Tensor, dirty: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """a4, incremental: re-normalise the rows flagged in `dirty` (int32 [N]) into the resident fp16 table and clear the flags."""
    N, D = mem.shape
    check(_lib.load().eod_memory_normalize_dirty_f16(mem.data_ptr(), obs.data_ptr(), dirty.data_ptr(), out.data_ptr(), N, D, _stream()),
          "eod_memory_normalize_dirty_f16")
    return out


def pooled_rows(H: int, W: int) -> int:
    """Rows of the pooled buffer: every level padded to whole 32-row operand tiles (include/eod_hip.h eod_memory_gather_pool)."""
    return int(_lib.load().eod_memory_pooled_halves(H, W)) // 512


def memory_gather_pool(mem_f16: torch.Tensor, proj: torch.Tensor, H: int, W: int, out: Optional[torch.Tensor] = None,
                       err: Optional[torch.Tensor] = None, torch_order: bool = False, batch: int = 1) -> torch.Tensor:
    """a8 gather + cascaded pooling -> fp16 pooled rows of the three levels in MFMA operand-fragment order (see the header).
    `batch` > 1: mem_f16 [B,N,D], proj [B,H,W], out [B * pooled_rows, D]."""
    N, D = mem_f16.shape[-2:]
    if out is None:
        out = torch.empty((batch * pooled_rows(H, W), D), dtype=torch.float16, device=mem_f16.device)
    check(_lib.load().eod_memory_gather_pool(mem_f16.data_ptr(), proj.data_ptr(), H, W, D, N, out.data_ptr(), _ptr(err), int(torch_order),
                                             batch, _stream()),
          "eod_memory_gather_pool")
    return out


class MemoryProjector:
    """`map_merge_projection{1,2,3}` + MAP_FEATURE_WEIGHT + fusion (timm.py:174-189) as one launch on the f16 matrix cores."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], device):
        self.lib = _lib.load()
        ws = [w.detach().to(torch.float32).reshape(256, 512).contiguous().to(device) for w in weights]
        bs = [b.detach().to(torch.float32).contiguous().to(device) for b in biases]
        self.prepared = torch.empty((self.lib.eod_memory_project_weights_bytes(),), dtype=torch.uint8, device=device)
        check(self.lib.eod_memory_project_prepare(ws[0].data_ptr(), bs[0].data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(), ws[2].data_ptr(),
                                                  bs[2].data_ptr(), self.prepared.data_ptr(), _stream()), "eod_memory_project_prepare")
        torch.cuda.current_stream().synchronize()       # ws / bs die with this frame; the prepare kernels must have read them

    def refresh(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]):
        """Re-prepare from stepped fp32 masters that live on the device ([256,512] / [256], contiguous, kept alive by the caller)."""
        _need_cuda(*weights, *biases)
        for w, b in zip(weights, biases):
            assert w.dtype == torch.float32 and w.is_contiguous() and w.numel() == 256 * 512 and b.is_contiguous() and b.numel() == 256
        check(self.lib.eod_memory_project_prepare(weights[0].data_ptr(), biases[0].data_ptr(), weights[1].data_ptr(), biases[1].data_ptr(),
                                                  weights[2].data_ptr(), biases[2].data_ptr(), self.prepared.data_ptr(), _stream()),
              "eod_memory_project_prepare")

    def __call__(self, pooled_f16: torch.Tensor, feats: torch.Tensor, H: int, W: int, weight: float, mode: str, batch: int = 1):
        """`batch` > 1: `feats` is level major over the scenes (include/eod_hip.h)."""
        _need_cuda(pooled_f16, feats)
        check(self.lib.eod_memory_project_fuse(pooled_f16.data_ptr(), self.prepared.data_ptr(), feats.data_ptr(), H, W, float(weight),
                                               {"sum": 0, "mem_only": 1}[mode], batch, _stream()), "eod_memory_project_fuse")
        return feats


class MemoryProjectorBackward:
    """Backward of `MemoryProjector` + the cascaded pools (timm.py:142-192): the memory-specific part of the training forward
    (first slice).  `__call__(grads, pooled_f16, H, W, weight)` with `grads` = dL/d(fused P3, P4, P5) as [P_l,256]
    fp32 rows -> dict(dW=[3 x [256,512]], db=[3 x [256]], gE=[3 x half [P_l,512]], gE2=fp32 [(H/4)(W/4),512])."""

    def __init__(self, weights: Sequence[torch.Tensor], device):
        self.lib = _lib.load()
        self.device = device
        # d(pooled) = weight * G . W: a 1x1 convolution whose [Cout=512][K=256] matrix is W^T
        self.convs = [Conv(w.detach().to(torch.float32).reshape(256, 512).t().contiguous().reshape(512, 256, 1, 1), None, device=device,
                           name=f"map_merge_projection{i + 1}^T") for i, w in enumerate(weights)]

    def refresh(self, weights: Sequence[torch.Tensor]) -> None:
        """After an optimizer step on the projections: the W^T convs of the input gradient follow the stepped weights."""
        for c, w in zip(self.convs, weights):
            c.w[:, :256].copy_(w.detach().reshape(256, 512).t())
            c.w_split = None

    def __call__(self, grads: Sequence[torch.Tensor], pooled_f16: torch.Tensor, H: int, W: int, weight: float,
                 need_input_grad: bool = True):
        """`need_input_grad=False`: only dW / db (the training step: the memory table is an input of `forward_model`, loader.py:199-223,
        nothing reads the gradient below the pooled operand -- three dgrad convs and the pool backward, ~50 MB of writes, stay out)."""
        _need_cuda(pooled_f16, *grads)
        dev = self.device
        rows = [(H >> (3 + l)) * (W >> (3 + l)) for l in range(3)]
        for g, r in zip(grads, rows):
            assert tuple(g.shape) == (r, 256) and g.dtype == torch.float32 and g.is_contiguous()
        dW = [torch.empty((256, 512), dtype=torch.float32, device=dev) for _ in range(3)]
        db = [torch.empty((256,), dtype=torch.float32, device=dev) for _ in range(3)]
        ws = _conv_ws.get(self.lib.eod_memory_project_backward_weights_workspace_bytes(), dev)      # the position ranges' partial sums
        check(self.lib.eod_memory_project_backward_weights_ws(grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(),
                                                              pooled_f16.data_ptr(), H, W, float(weight), dW[0].data_ptr(), db[0].data_ptr(),
                                                              dW[1].data_ptr(), db[1].data_ptr(), dW[2].data_ptr(), db[2].data_ptr(),
                                                              ws.data_ptr(), ws.numel(), _stream()),
              "eod_memory_project_backward_weights_ws")
        if not need_input_grad:
            return dict(dW=dW, db=db)
        dec = [self.convs[l](grads[l].view(1, H >> (3 + l), W >> (3 + l), 256), 1, H >> (3 + l), W >> (3 + l), out_scale=float(weight))
               .view(rows[l], 512) for l in range(3)]
        gE = [torch.empty((rows[l], 512), dtype=torch.float16, device=dev) for l in range(3)]
        gE2 = torch.empty(((H >> 2) * (W >> 2), 512), dtype=torch.float32, device=dev)
        check(self.lib.eod_memory_pool_backward(dec[0].data_ptr(), dec[1].data_ptr(), dec[2].data_ptr(), H, W, gE[0].data_ptr(),
                                                gE[1].data_ptr(), gE[2].data_ptr(), gE2.data_ptr(), _stream()), "eod_memory_pool_backward")
        return dict(dW=dW, db=db, gE=gE, gE2=gE2, dEc=dec)


class ConvBackward:
    """Backward of a `Conv` layer with bias and optional ReLU (third slice): stride-1 'same' layers (CenterNet
    tower, FPN output convs, mask head convs, the trunk's 1x1 / 3x3 convs), strided ones (P6 / P7, the trunk's down-sampling convs)
    and the weight gradient of the 4-channel stem.  `__call__(x, y, g_out)` with the forward's input `x`
    [N,H,W,Cin], output `y` [N,H,W,Cout] (post-ReLU, only read when `relu`) and dL/dy -> dict(dx, dw [Cout, KH*KW*Cin] in the packed
    layout of `Conv.w`, db).  dx = `eod_conv2d` of the pre-activation gradient with the 180-degree rotated, in/out-transposed
    weights; dw / db = `eod_conv2d_backward_weights`."""

    _workspace: Dict = {}      # per device: partial sums of the position-split weight gradient (stream-ordered reuse)
    # The weight gradients are side outputs of the backward chain (only the optimizer reads them), the input gradients ARE the chain:
    # with `side_stream` every dW / db launch goes to a second stream behind an event of the main one and runs beside the next
    # layers' dgrad convs (the training step is one stream of ~30 us kernels otherwise).  Whoever reads a dW on the main stream calls
    # `join(device)` first -- or does its own small ops inside `on_side(device)`.  Per instance (`ConvBackward(conv, side_stream=True)`):
    # off by default, `modeling.training` switches it on for the layers of its steps.
    _side: Dict = {}           # per device: [stream, event recorded behind the last side launch]

    @classmethod
    def _side_state(cls, dev):
        key = torch.device(dev).index or 0
        st = cls._side.get(key)
        if st is None:
            st = cls._side[key] = [torch.cuda.Stream(device=dev), None]
        return st

    @classmethod
    def join(cls, dev) -> None:
        """The current stream waits for every weight-gradient launch (and `on_side` op) issued so far."""
        st = cls._side.get(torch.device(dev).index or 0)
        if st is not None and st[1] is not None:
            torch.cuda.current_stream(dev).wait_event(st[1])

    @classmethod
    def on_side(cls, dev, enabled: bool = True):
        """Context: torch ops on weight gradients (slices, level sums) ordered on the side stream, behind what is there already."""
        import contextlib
        if not enabled:
            return contextlib.nullcontext()
        st = cls._side_state(dev)

        @contextlib.contextmanager
        def ctx():
            with torch.cuda.stream(st[0]):
                yield
                ev = torch.cuda.Event()
                ev.record(st[0])
                st[1] = ev
        return ctx()

    def __init__(self, conv: "Conv", side_stream: bool = False):
        if conv.out_mode != 0:
            raise ValueError("ConvBackward covers plain convolutions (no deconv)")
        self.side_stream = bool(side_stream)
        # stride-1 'same' layers: dX on the matrix cores (eod_conv2d with rotated weights); anything else: the gather kernel
        self.same = conv.stride == 1 and conv.KH == conv.KW and conv.pad * 2 == conv.KH - 1
        # strided 'same'-padded layers (3x3 s2 p1: P6 / P7 and the trunk's conv2; 1x1 s2: its downsample convs): the gradient is
        # spread onto the input grid with zeros in between and goes through the same rotated-weights conv on the matrix cores
        # (3/4 of the products are zeros, still ~20x faster than the gather kernel: 0.87 ms -> tens of us per layer at 640x640)
        self.zero_insert = conv.stride > 1 and conv.KH == conv.KW and conv.pad * 2 == conv.KH - 1
        self.conv = conv
        self.lib = _lib.load()
        self._flipped = None
        self._flipped_of = None

    def _dgrad_conv(self) -> "Conv":
        # rebuilt when the forward weights have been replaced / updated in place (an optimizer step): keyed by the tensor's version
        key = (self.conv.w.data_ptr(), self.conv.w._version)
        if self._flipped is None or self._flipped_of != key:
            c = self.conv
            K = c.KH * c.KW * c.Cin
            w = c.w[:, :K].reshape(c.Cout, c.KH, c.KW, c.Cin)                       # [co, ky, kx, ci]
            if self._flipped is None:
                wt = w.flip(1, 2).permute(3, 0, 1, 2).contiguous()                 # [ci, co, ky', kx'] = OIHW of the transposed conv
                self._flipped = Conv(wt.cpu(), None, stride=1, pad=c.pad, device=c.w.device, name=c.name + "^T")
            else:
                # the weights were stepped: refresh the packed rows [ci][(ky', kx', co)] on the device, one launch
                check(self.lib.eod_conv_rotate_weights(c.w.data_ptr(), c.Cout, c.KH, c.KW, c.Cin, c.Kpad, self._flipped.w.data_ptr(),
                                                       self._flipped.Kpad, _stream()), "eod_conv_rotate_weights")
                self._flipped.w_split = None
            self._flipped_of = key
        return self._flipped

    def _pyramid(self, x, y, g_out, relu, need_dx, dx_res, dx_gate, levels):
        """`levels=(row_offsets, [(h, w), ...])`: a level-shared layer over a feature pyramid stored as one row list (x [rows, Cin],
        g_out [rows, Cout]; the forward's `Conv(..., levels=)`): dW / db summed over the levels by ONE weight-gradient launch
        (`eod_conv2d_backward_weights_levels`) and dX by one pyramid-mode launch of the rotated-weights conv."""
        c = self.conv
        if not self.same:
            raise ValueError("pyramid mode covers stride-1 'same' layers")
        off, shapes = levels
        L, rows = len(shapes), int(off[-1])
        assert tuple(x.shape) == (rows, c.Cin) and tuple(g_out.shape) == (rows, c.Cout) and x.is_contiguous() and g_out.is_contiguous()
        g = g_out
        if relu:
            g = torch.empty_like(g_out)
            check(self.lib.eod_relu_backward(g_out.data_ptr(), y.data_ptr(), g.data_ptr(), g.numel(), _stream()), "eod_relu_backward")
        lo = (C.c_int32 * (L + 1))(*off)
        lh = (C.c_int32 * L)(*[h for h, _ in shapes])
        lw = (C.c_int32 * L)(*[w for _, w in shapes])
        K = c.KH * c.KW * c.Cin
        dw = torch.empty((c.Cout, K), dtype=torch.float32, device=x.device)
        db = torch.empty((c.Cout,), dtype=torch.float32, device=x.device)
        need = self.lib.eod_conv2d_backward_weights_levels_workspace_bytes(rows, c.Cin, c.Cout, c.KH, c.KW)
        ws = ConvBackward._workspace.get(x.device)
        if need and (ws is None or ws.numel() * 4 < need):
            ws = ConvBackward._workspace[x.device] = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=x.device)
        check(self.lib.eod_conv2d_backward_weights_levels(x.data_ptr(), g.data_ptr(), L, lo, lh, lw, c.Cin, c.Cout, c.KH, c.KW, c.pad,
                                                          dw.data_ptr(), db.data_ptr(), ws.data_ptr() if need else None,
                                                          ws.numel() * 4 if need else 0, _stream()), "eod_conv2d_backward_weights_levels")
        dx = None
        if need_dx:
            dx = self._dgrad_conv()(g, 1, 0, 0, levels=(off, shapes), res=dx_res, res_mode=1 if dx_res is not None else 0, gate=dx_gate)
        return dict(dx=dx, dw=dw, db=db)

    @staticmethod
    def refresh_all(bws: Sequence["ConvBackward"]) -> None:
        """After an optimizer step: the rotated weights of every layer in `bws` that has an input-gradient convolution, in
        ceil(n / 24) launches (`eod_conv_rotate_weights_multi`) instead of one launch per layer at its next backward."""
        todo = [bw for bw in bws if bw._flipped is not None]
        for bw in bws:
            bw._flipped_of = None
        if not todo:
            return
        descs = (_lib.EodRotateTensor * len(todo))()
        for d, bw in zip(descs, todo):
            c = bw.conv
            d.w, d.out = c.w.data_ptr(), bw._flipped.w.data_ptr()
            d.Cout, d.KH, d.KW, d.Cin, d.ld_in, d.ld_out = c.Cout, c.KH, c.KW, c.Cin, c.Kpad, bw._flipped.Kpad
        check(_lib.load().eod_conv_rotate_weights_multi(descs, len(todo), _stream()), "eod_conv_rotate_weights_multi")
        for bw in todo:
            bw._flipped.w_split = None
            bw._flipped_of = (bw.conv.w.data_ptr(), bw.conv.w._version)

    def __call__(self, x: torch.Tensor, y: Optional[torch.Tensor], g_out: torch.Tensor, relu: bool = False, need_dx: bool = True,
                 dx_res: Optional[torch.Tensor] = None, dx_gate: Optional[torch.Tensor] = None,
                 levels: Optional[Tuple[Sequence[int], Sequence[Tuple[int, int]]]] = None):
        c = self.conv
        _need_cuda(x, y, g_out)
        if levels is not None:
            return self._pyramid(x, y, g_out, relu, need_dx, dx_res, dx_gate, levels)
        if c.tap4 and need_dx:
            raise ValueError("the 4-channel stem has a weight gradient only (its input is the image): pass need_dx=False")
        N, H, W, _ = x.shape
        OH, OW = c.out_hw(H, W)
        assert tuple(g_out.shape) == (N, OH, OW, c.Cout) and x.is_contiguous() and g_out.is_contiguous()
        g = g_out
        if relu:
            g = torch.empty_like(g_out)
            check(self.lib.eod_relu_backward(g_out.data_ptr(), y.data_ptr(), g.data_ptr(), g.numel(), _stream()), "eod_relu_backward")
        K = c.KH * c.KW * c.Cin
        need = self.lib.eod_conv2d_backward_weights_workspace_bytes(N, H, W, c.Cin, c.Cout, c.KH, c.KW, c.pad, c.stride)

        def wgrad():
            dw = torch.empty((c.Cout, K), dtype=torch.float32, device=x.device)
            db = torch.empty((c.Cout,), dtype=torch.float32, device=x.device)
            ws = ConvBackward._workspace.get(x.device)
            if need and (ws is None or ws.numel() * 4 < need):
                ws = ConvBackward._workspace[x.device] = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=x.device)
            check(self.lib.eod_conv2d_backward_weights_ws(x.data_ptr(), g.data_ptr(), N, H, W, c.Cin, c.Cout, c.KH, c.KW, c.pad, c.stride,
                                                          dw.data_ptr(), db.data_ptr(), ws.data_ptr() if need else None,
                                                          ws.numel() * 4 if need else 0, _stream()), "eod_conv2d_backward_weights_ws")
            return dw, db

        if self.side_stream:
            main = torch.cuda.current_stream(x.device)
            st = ConvBackward._side_state(x.device)
            ready = torch.cuda.Event()
            ready.record(main)                           # x and g (the ReLU-masked gradient included) are complete in main's order here
            st[0].wait_event(ready)
            with torch.cuda.stream(st[0]):
                dw, db = wgrad()
                done = torch.cuda.Event()
                done.record(st[0])
                st[1] = done
            for t in (x, g):
                t.record_stream(st[0])                   # the caching allocator must not hand their memory out before the side launch ran
            dw.record_stream(main)
            db.record_stream(main)
        else:
            dw, db = wgrad()
        # `dx_res` (a second gradient of x: the skip connection's) and `dx_gate` (the ReLU output x came out of: its backward) ride
        # on the input-gradient convolution's epilogue: dx = relu'(dx_gate) * (conv(g) + dx_res) in one launch
        dx = None
        tail = dict(res=dx_res, res_mode=1 if dx_res is not None else 0, gate=dx_gate)
        if need_dx and self.same:
            dx = self._dgrad_conv()(g, N, H, W, **tail)
        elif need_dx and self.zero_insert:
            up = torch.zeros((N, H, W, c.Cout), dtype=torch.float32, device=x.device)
            up[:, ::c.stride, ::c.stride] = g
            dx = self._dgrad_conv()(up, N, H, W, **tail)
        elif need_dx:
            dx = torch.empty_like(x)
            check(self.lib.eod_conv2d_backward_input(g.data_ptr(), c.w.data_ptr(), c.Kpad, N, H, W, c.Cin, c.Cout, c.KH, c.KW, c.pad, c.stride,
                                                     dx.data_ptr(), _stream()), "eod_conv2d_backward_input")
            if dx_res is not None:
                dx = dx + dx_res
            if dx_gate is not None:
                gated = torch.empty_like(dx)
                check(self.lib.eod_relu_backward(dx.data_ptr(), dx_gate.data_ptr(), gated.data_ptr(), dx.numel(), _stream()), "eod_relu_backward")
                dx = gated
        return dict(dx=dx, dw=dw, db=db)


CENTERNET_SOI = ((0, 80), (64, 160), (128, 320), (256, 640), (512, 10000000))      # MODEL.CENTERNET.SOI


def centernet_targets(gt_boxes: torch.Tensor, level_hw: Sequence[Tuple[int, int]], strides: Sequence[int] = (8, 16, 32, 64, 128),
                      sizes_of_interest=CENTERNET_SOI, hm_min_overlap: float = 0.8, min_radius: float = 4.0):
    """CenterNet's target assignment for one image on the device (`eod_centernet_targets`; centernet.py:342-479) ->
    (agn_heatmap [P], reg_targets [P,4], pos_inds int32 [N * levels] of which counts[0] are valid, counts int32 [2] = positives,
    regression rows).  gt_boxes [N,4] float32 on the device."""
    _need_cuda(gt_boxes)
    dev = gt_boxes.device
    d = _lib.EodCenterNetTargetDesc()
    N, L = gt_boxes.shape[0], len(level_hw)
    off = [0]
    for (h, w) in level_hw:
        off.append(off[-1] + h * w)
    P = off[-1]
    d.gt_boxes, d.n_boxes, d.levels = (gt_boxes.data_ptr() if N else None), N, L
    for i, v in enumerate(off):
        d.level_off[i] = v
    for l in range(L):
        d.level_w[l], d.level_stride[l] = level_hw[l][1], strides[l]
        d.soi_lo[l], d.soi_hi[l] = float(sizes_of_interest[l][0]), float(sizes_of_interest[l][1])
    d.hm_min_overlap, d.min_radius = hm_min_overlap, min_radius
    heat = torch.empty((P,), dtype=torch.float32, device=dev)
    reg = torch.empty((P, 4), dtype=torch.float32, device=dev)
    pos = torch.zeros((max(N * L, 1),), dtype=torch.int32, device=dev)
    counts = torch.zeros((2,), dtype=torch.int32, device=dev)
    d.agn_heatmap, d.reg_targets, d.pos_inds, d.counts = heat.data_ptr(), reg.data_ptr(), pos.data_ptr(), counts.data_ptr()
    check(_lib.load().eod_centernet_targets(C.byref(d), _stream()), "eod_centernet_targets")
    return heat, reg, pos, counts


class CenterNetLoss:
    """`CenterNet.losses` of the recurrent configuration on the device with its gradient (`eod_centernet_loss`; centernet.py:241-318:
    agnostic heatmap focal loss + GIoU regression loss).  `__call__(head_out [P, stride], agn_heatmap [P], reg_targets [P,4],
    pos_inds int32 [N], num_pos_avg, reg_norm)` -> (losses [3] = loc, agn_pos, agn_neg on the device, dL/d(head_out)); the two norms are
    the all-reduced counts / world size the reference divides by (the caller reduces them over the ranks)."""

    def __init__(self, level_off: Sequence[int], level_scale: Sequence[float], device, head_stride: int = 8, alpha: float = 0.25,
                 beta: float = 4.0, gamma: float = 2.0, sigmoid_clamp: float = 1e-4, ignore_high_fp: float = 0.85,
                 pos_weight: float = 0.5, neg_weight: float = 0.5, reg_weight: float = 1.0):
        self.lib = _lib.load()
        d = _lib.EodCenterNetLossDesc()
        L = len(level_scale)
        assert len(level_off) == L + 1 and L <= 8
        d.head_stride, d.P, d.levels = head_stride, level_off[-1], L
        for i, v in enumerate(level_off):
            d.level_off[i] = v
        for i, v in enumerate(level_scale):
            d.level_scale[i] = float(v)
        d.hm_focal_alpha, d.hm_focal_beta, d.loss_gamma, d.sigmoid_clamp, d.ignore_high_fp = alpha, beta, gamma, sigmoid_clamp, ignore_high_fp
        d.pos_weight, d.neg_weight, d.reg_weight = pos_weight, neg_weight, reg_weight
        nbytes = self.lib.eod_centernet_loss_workspace_bytes()
        self.ws = torch.empty((nbytes // 8,), dtype=torch.float64, device=device)
        d.workspace, d.workspace_bytes = self.ws.data_ptr(), nbytes
        self.losses = torch.zeros((3,), dtype=torch.float32, device=device)
        d.losses = self.losses.data_ptr()
        self.desc = d

    def __call__(self, head_out: torch.Tensor, agn_heatmap: torch.Tensor, reg_targets: torch.Tensor, pos_inds: torch.Tensor,
                 num_pos_avg: float = 1.0, reg_norm: float = 1.0, out: Optional[torch.Tensor] = None, counts_local: Optional[torch.Tensor] = None,
                 counts_total: Optional[torch.Tensor] = None, world_size: int = 1):
        """`counts_local` / `counts_total` (int32 [2] on the device: `centernet_targets`' counts, the second all-reduced over the
        ranks): the positives' list length and the two normalisers are read on the device -- `pos_inds` is then the capacity-sized
        list and `num_pos_avg` / `reg_norm` are not used; no host round trip between the target assignment and the losses."""
        _need_cuda(head_out, agn_heatmap, reg_targets, pos_inds, out, counts_local, counts_total)
        d = self.desc
        assert tuple(head_out.shape) == (d.P, d.head_stride) and head_out.is_contiguous() and pos_inds.dtype == torch.int32
        assert tuple(reg_targets.shape) == (d.P, 4) and reg_targets.is_contiguous() and agn_heatmap.numel() == d.P
        if out is None:
            out = torch.empty_like(head_out)
        d.head_out, d.agn_heatmap, d.reg_targets = head_out.data_ptr(), agn_heatmap.data_ptr(), reg_targets.data_ptr()
        d.pos_inds, d.n_pos = (pos_inds.data_ptr() if pos_inds.numel() else None), pos_inds.numel()
        d.num_pos_avg, d.reg_norm, d.d_head_out = float(num_pos_avg), float(reg_norm), out.data_ptr()
        d.counts_local, d.counts_total, d.world_size = _ptr(counts_local), _ptr(counts_total), float(world_size)
        check(self.lib.eod_centernet_loss(C.byref(d), _stream()), "eod_centernet_loss")
        return self.losses, out


def fast_rcnn_loss(scores: torch.Tensor, deltas: torch.Tensor, proposal_boxes: torch.Tensor, gt_boxes: torch.Tensor,
                   gt_classes: torch.Tensor, num_classes: int, box_weights: Sequence[float], class_weight: Optional[torch.Tensor] = None,
                   smooth_l1_beta: float = 0.0):
    """One cascade stage's `DeticFastRCNNOutputLayers.losses` (sigmoid CE + class-agnostic smooth-L1, detic_fast_rcnn.py:157-303) on the
    device -> (losses [2] = loss_cls, loss_box_reg; dL/d(scores) [B, ld]; dL/d(deltas) [B,4]).  gt_classes int32, background = C."""
    _need_cuda(scores, deltas, proposal_boxes, gt_boxes, gt_classes, class_weight)
    B, ld = scores.shape
    assert scores.is_contiguous() and tuple(deltas.shape) == (B, 4) and deltas.is_contiguous() and gt_classes.dtype == torch.int32
    lib = _lib.load()
    nbytes = lib.eod_fast_rcnn_loss_workspace_bytes(B)
    ws = torch.empty((nbytes // 8,), dtype=torch.float64, device=scores.device)
    losses = torch.empty((2,), dtype=torch.float32, device=scores.device)
    ds, dd = torch.empty_like(scores), torch.empty_like(deltas)
    wx, wy, ww, wh = box_weights
    check(lib.eod_fast_rcnn_loss(scores.data_ptr(), ld, deltas.data_ptr(), proposal_boxes.data_ptr(), gt_boxes.data_ptr(),
                                 gt_classes.data_ptr(), _ptr(class_weight), B, num_classes, wx, wy, ww, wh, smooth_l1_beta, ds.data_ptr(),
                                 dd.data_ptr(), losses.data_ptr(), ws.data_ptr(), nbytes, _stream()), "eod_fast_rcnn_loss")
    return losses, ds, dd


def match_label(boxes: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor, iou_thresh: float, num_classes: int):
    """detectron2's pairwise_iou + Matcher([iou_thresh], [0, 1]) + the labelling of `_sample_proposals` / `_match_and_label_boxes`
    (called at detic_roi_heads.py:232,115) -> (matched_idx int32 [R], matched_iou [R], classes int32 [R] with background =
    num_classes, matched gt boxes [R,4]).  gt_boxes [G,4] / gt_classes int32 [G]; G may be 0."""
    _need_cuda(boxes, gt_boxes, gt_classes)
    R, G = int(boxes.shape[0]), int(gt_boxes.shape[0])
    assert boxes.is_contiguous() and boxes.dtype == torch.float32 and tuple(boxes.shape) == (R, 4)
    assert gt_boxes.is_contiguous() and gt_boxes.dtype == torch.float32 and gt_classes.dtype == torch.int32 and gt_classes.numel() == G
    dev = boxes.device
    midx = torch.empty((R,), dtype=torch.int32, device=dev)
    miou = torch.empty((R,), dtype=torch.float32, device=dev)
    cls = torch.empty((R,), dtype=torch.int32, device=dev)
    gtb = torch.empty((R, 4), dtype=torch.float32, device=dev)
    check(_lib.load().eod_match_label(boxes.data_ptr(), R, gt_boxes.data_ptr() if G else None, gt_classes.data_ptr() if G else None, G,
                                      float(iou_thresh), num_classes, midx.data_ptr(), miou.data_ptr(), cls.data_ptr(), gtb.data_ptr(),
                                      _stream()), "eod_match_label")
    return midx, miou, cls, gtb


def match_label_proposals(prop_boxes: torch.Tensor, prop_count: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor,
                          iou_thresh: float, num_classes: int, append_gt: bool = True):
    """`match_label` on a capacity-sized proposal list whose length is on the device, the ground truth appended behind the live rows
    (detectron2's add_ground_truth_to_proposals) -> (all_boxes [cap + G, 4], classes int32 [cap + G] with -1 = no row, matched gt
    boxes): nothing is read back."""
    _need_cuda(prop_boxes, prop_count, gt_boxes, gt_classes)
    cap, G = int(prop_boxes.shape[0]), int(gt_boxes.shape[0])
    assert prop_boxes.is_contiguous() and prop_boxes.dtype == torch.float32 and prop_count.dtype == torch.int32
    assert gt_boxes.is_contiguous() and gt_boxes.dtype == torch.float32 and gt_classes.dtype == torch.int32 and gt_classes.numel() == G
    dev = prop_boxes.device
    R = cap + (G if append_gt else 0)
    allb = torch.empty((R, 4), dtype=torch.float32, device=dev)
    midx = torch.empty((R,), dtype=torch.int32, device=dev)
    miou = torch.empty((R,), dtype=torch.float32, device=dev)
    cls = torch.empty((R,), dtype=torch.int32, device=dev)
    gtb = torch.empty((R, 4), dtype=torch.float32, device=dev)
    check(_lib.load().eod_match_label_proposals(prop_boxes.data_ptr(), prop_count.data_ptr(), cap, gt_boxes.data_ptr() if G else None,
                                                gt_classes.data_ptr() if G else None, G, int(append_gt), float(iou_thresh), num_classes,
                                                allb.data_ptr(), midx.data_ptr(), miou.data_ptr(), cls.data_ptr(), gtb.data_ptr(), _stream()),
          "eod_match_label_proposals")
    return allb, cls, gtb


def sample_proposals(classes: torch.Tensor, keys: torch.Tensor, num_classes: int, batch_size_per_image: int, positive_fraction: float):
    """detectron2's `subsample_labels` as a selection by random keys (include/eod_hip.h) -> (sampled_idx int32 [batch], counts int32
    [2] = foreground rows, rows), both on the device."""
    _need_cuda(classes, keys)
    R = int(classes.numel())
    assert classes.dtype == torch.int32 and keys.dtype == torch.float32 and keys.numel() == R
    idx = torch.zeros((batch_size_per_image,), dtype=torch.int32, device=classes.device)
    counts = torch.zeros((2,), dtype=torch.int32, device=classes.device)
    # the ABI takes the fraction as a float; hand over the mid-point of the bucket of detectron2's `int(batch * fraction)` (a double
    # product) so that the float product truncates to the same count for every fraction (0.7 x 10 is 7 in double and 6.9999999 in float)
    max_pos = int(batch_size_per_image * positive_fraction)
    fraction = min((max_pos + 0.5) / batch_size_per_image, 1.0)
    st = _lib.load().eod_sample_proposals(classes.data_ptr(), keys.data_ptr(), R, num_classes, batch_size_per_image,
                                          float(fraction), idx.data_ptr(), counts.data_ptr(), _stream())
    if st == -5:
        raise _lib.EodError(f"eod_sample_proposals: {R} proposals exceed the sampling kernel's capacity of 8192 rows")
    check(st, "eod_sample_proposals")
    return idx, counts


def zs_logits(feat: torch.Tensor, zs: torch.Tensor, temp: float = 50.0, ld: Optional[int] = None, featn_out: Optional[torch.Tensor] = None):
    """The training-mode scores of DeticFastRCNNOutputLayers.forward: temp * normalize(feat [B,512]) @ zs [512,C1] -> logits [B, ld]."""
    _need_cuda(feat, zs, featn_out)
    B, C1 = int(feat.shape[0]), int(zs.shape[1])
    feat = feat.view(B, 512)
    assert feat.is_contiguous() and zs.is_contiguous() and int(zs.shape[0]) == 512
    ld = C1 if ld is None else ld
    out = torch.zeros((B, ld), dtype=torch.float32, device=feat.device)
    check(_lib.load().eod_zs_logits(feat.data_ptr(), zs.data_ptr(), B, 512, C1, float(temp), out.data_ptr(), ld, _ptr(featn_out),
                                    _stream()), "eod_zs_logits")
    return out


def zs_logits_backward(feat: torch.Tensor, zs: torch.Tensor, d_logits: torch.Tensor, temp: float = 50.0) -> torch.Tensor:
    """d feat [B,512] of `zs_logits` given d_logits [B, ld] (`eod_zs_logits_backward`)."""
    _need_cuda(feat, zs, d_logits)
    B, C1 = int(d_logits.shape[0]), int(zs.shape[1])
    feat = feat.view(B, 512)
    assert feat.is_contiguous() and d_logits.is_contiguous() and d_logits.shape[1] >= C1
    out = torch.empty((B, 512), dtype=torch.float32, device=feat.device)
    check(_lib.load().eod_zs_logits_backward(feat.data_ptr(), zs.data_ptr(), d_logits.data_ptr(), int(d_logits.shape[1]), B, 512, C1,
                                             float(temp), out.data_ptr(), _stream()), "eod_zs_logits_backward")
    return out


class AdamW:
    """`torch.optim.AdamW` (single-tensor form) + detectron2's clip-by-value on the device, one `eod_adamw_step` launch per parameter
    tensor: the optimizer of the reference's training configuration (custom_solver.py:69-72, Base-...recurrent.yaml:68-74).
    `groups`: what `solver.build_param_groups` returns (each with its own `lr`); state = exp_avg / exp_avg_sq per tensor."""

    def __init__(self, groups, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, clip_value: float = 0.0):
        self.groups = groups
        self.betas, self.eps, self.weight_decay, self.clip_value = betas, eps, weight_decay, clip_value
        self.state = [(torch.zeros_like(g["param"]), torch.zeros_like(g["param"])) for g in groups]
        self.steps = [0] * len(groups)
        self.lib = _lib.load()
        # all tensors in ceil(n / 20) launches (`eod_adamw_step_multi`); False: one `eod_adamw_step` launch per tensor (the same
        # arithmetic element for element: tests/test_backward_gpu.py compares the two)
        self.multi_tensor = True
        self._descs = (_lib.EodAdamWTensor * max(len(groups), 1))()

    def state_dict(self) -> Dict:
        """{parameter name: {'step', 'exp_avg', 'exp_avg_sq'}} on the host -- what DetectionCheckpointer stores for the optimizer
        (train_mp3d.py:521-523), keyed by the reference's parameter names instead of torch's positional ids."""
        return {g["name"]: {"step": self.steps[i], "exp_avg": self.state[i][0].cpu(), "exp_avg_sq": self.state[i][1].cpu()}
                for i, g in enumerate(self.groups)}

    def load_state_dict(self, sd: Dict) -> None:
        missing = [g["name"] for g in self.groups if g["name"] not in sd]
        if missing:
            raise KeyError(f"optimizer state without {len(missing)} of this model's parameters (first: {missing[0]})")
        for i, g in enumerate(self.groups):
            e = sd[g["name"]]
            if tuple(e["exp_avg"].shape) != tuple(self.state[i][0].shape):
                raise ValueError(f"optimizer state of {g['name']}: shape {tuple(e['exp_avg'].shape)} != {tuple(self.state[i][0].shape)}")
            self.steps[i] = int(e["step"])
            self.state[i][0].copy_(e["exp_avg"])
            self.state[i][1].copy_(e["exp_avg_sq"])

    def step(self, grads: Sequence[Optional[torch.Tensor]], lr_factor: float = 1.0):
        """`grads[i]`: gradient of group i's tensor (None: no gradient this iteration, the tensor is skipped as torch does)."""
        if self.multi_tensor:
            n = 0
            for i, (g, grad) in enumerate(zip(self.groups, grads)):
                if grad is None:
                    continue
                p = g["param"]
                _need_cuda(p, grad)
                assert p.dtype == torch.float32 and grad.dtype == torch.float32 and p.is_contiguous() and grad.is_contiguous()
                assert grad.shape == p.shape
                self.steps[i] += 1
                d = self._descs[n]
                d.param, d.grad, d.exp_avg, d.exp_avg_sq = p.data_ptr(), grad.data_ptr(), self.state[i][0].data_ptr(), self.state[i][1].data_ptr()
                d.n, d.lr, d.weight_decay, d.step = p.numel(), g["lr"] * lr_factor, g.get("weight_decay", self.weight_decay), self.steps[i]
                fold = g.get("fold")              # (folded weights [rows, ld], per-row scale [rows]): written by the same launch
                if fold is not None:
                    d.folded_out, d.row_scale, d.cols, d.ld_out = fold[0].data_ptr(), fold[1].data_ptr(), p.shape[1], fold[0].stride(0)
                    # `grads[i]` is the gradient of the folded weights: x scale inside the launch (group key "grad_of_folded")
                    d.grad_of_folded = 1 if g.get("grad_of_folded") else 0
                else:
                    d.folded_out, d.row_scale, d.cols, d.ld_out, d.grad_of_folded = None, None, 0, 0, 0
                n += 1
            if n:
                check(self.lib.eod_adamw_step_multi(self._descs, n, self.betas[0], self.betas[1], self.eps, self.clip_value, _stream()),
                      "eod_adamw_step_multi")
            return
        for i, (g, grad) in enumerate(zip(self.groups, grads)):
            if grad is None:
                continue
            p = g["param"]
            _need_cuda(p, grad)
            assert p.dtype == torch.float32 and grad.dtype == torch.float32 and p.is_contiguous() and grad.is_contiguous()
            assert grad.shape == p.shape
            self.steps[i] += 1
            m, v = self.state[i]
            if g.get("fold") is not None and g.get("grad_of_folded"):
                grad = grad * g["fold"][1].view(-1, 1)
            check(self.lib.eod_adamw_step(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), g["lr"] * lr_factor,
                                          self.betas[0], self.betas[1], self.eps, g.get("weight_decay", self.weight_decay), self.steps[i],
                                          self.clip_value, _stream()), "eod_adamw_step")
            if g.get("fold") is not None:
                g["fold"][0][:, :p.shape[1]].copy_(p * g["fold"][1].view(-1, 1))


class MemoryWriter:
    """a16-a19 write path (custom_rcnn.py:681-760) on device."""

    def __init__(self, H: int, W: int, n_cells: int, K_cap: int, R_cap: int, device, mask_thresh: float = 0.5, batch: int = 1):
        """`batch` > 1: B scenes with B independent states; every buffer handed to the call is B single-scene buffers back to back."""
        self.lib = _lib.load()
        nbytes = self.lib.eod_memory_write_workspace_bytes(H, W, 512, n_cells, K_cap, R_cap) * batch
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        check(self.lib.eod_memory_write_init(self.ws.data_ptr(), nbytes, H, W, 512, n_cells, K_cap, R_cap, _stream()), "eod_memory_write_init")
        self.k_out = torch.zeros((batch,), dtype=torch.int32, device=device)
        d = EodMemWriteDesc()
        d.batch = batch
        d.K_cap, d.R_cap, d.H, d.W, d.D, d.n_cells, d.mask_thresh = K_cap, R_cap, H, W, 512, n_cells, mask_thresh
        d.workspace, d.workspace_bytes, d.k_out = self.ws.data_ptr(), nbytes, self.k_out.data_ptr()
        self.desc = d

    def __call__(self, featn, prop_boxes, prop_masks, det_rows, det_count, proj, mem, obs, dirty=None, err=None, snapshot=None):
        """`snapshot` (fp16 [N,512], the table `memory_gather_pool` reads): refreshed in place for every cell whose observation
        count changes (then `dirty` is not written); `dirty` alone: those cells are only marked for `memory_normalize_dirty_f16`."""
        d = self.desc
        d.dirty, d.err_flags, d.snapshot_f16 = _ptr(dirty), _ptr(err), _ptr(snapshot)
        d.featn, d.prop_boxes, d.prop_masks = featn.data_ptr(), prop_boxes.data_ptr(), prop_masks.data_ptr()
        d.det_rows, d.det_count, d.proj, d.mem, d.obs = det_rows.data_ptr(), det_count.data_ptr(), proj.data_ptr(), mem.data_ptr(), obs.data_ptr()
        check(self.lib.eod_memory_write(C.byref(d), _stream()), "eod_memory_write")
        return self.k_out


def semmap_labels(mem: torch.Tensor, obs: torch.Tensor, zs: torch.Tensor, thresh: float) -> torch.Tensor:
    """a20 (custom_rcnn.py:745-756,938-1017): int32 [N] labels, -1 below the observation-intensity threshold."""
    N, D = mem.shape
    labels = torch.empty((N,), dtype=torch.int32, device=mem.device)
    ws = torch.empty((N + 4,), dtype=torch.float32, device=mem.device)
    check(_lib.load().eod_semmap_labels(mem.data_ptr(), obs.data_ptr(), zs.data_ptr(), N, D, zs.shape[1], thresh, labels.data_ptr(),
                                        ws.data_ptr(), _stream()), "eod_semmap_labels")
    return labels
