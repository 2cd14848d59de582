"""B independent sequences in lock-step on one GPU with N = B through EVERY stage of the frame (BASELINE.json configs[4]: 4 sequences
at 960x960, 512x512 memory grid): one launch per stage for all scenes, B independent memory states.

Only independent sequences may be batched: inside a sequence frame t+1 reads the memory frame t wrote
(`Detic/detic/modeling/meta_arch/custom_rcnn.py:485-515`, `Detic/SMNet/loader.py:289-293`).

The frame is the one `CustomRCNNRecurrent.inference_frame` runs (`custom_rcnn.py:548-582` + `update_implicit_memory` 681-760), stage
by stage, on buffers that hold the B scenes back to back (the batch convention of include/eod_hip.h):

  preprocess (B small launches: the images arrive as B tensors) -> ResNet-50 trunk + FPN top-down, N = B (`timm.py:277-299,118-136`)
  -> memory read: gather + cascaded pooling and projection + fusion, grid.y = scene, B fp16 tables (`timm.py:142-192`)
  -> P6 / P7, N = B -> CenterNet tower + GroupNorm on the 5 B level images of the batch as one row list (`centernet_head.py:141-161`)
  -> proposal decoding, one workgroup per (level, scene) and per scene (`centernet.py:603-745`)
  -> cascade: ROIAlign over B x R boxes, the FC layers as ONE GEMM over B x R rows (per-scene counts: EodConvDesc.m_segments),
     classifier / deltas over B x R rows, detection selection one workgroup per scene (`detic_roi_heads.py:88-222`)
  -> memory selection, one workgroup per scene (`custom_rcnn.py:825-875`)
  -> mask head on the memory instances and on the detection groups of ALL scenes: the scenes' lists are concatenated
     (eod_concat_lists) and each pass is one ROIAlign + 4 convs + the fused tail over the concatenation (`detic_roi_heads.py:257-268`)
  -> memory write: three launches for the B states (`custom_rcnn.py:884-936`) -> post-processing + paste, grid = scene.

Every layer is planned like ONE scene (`plan_rows`), so it walks K exactly as the single-scene call: the results of every scene
are bitwise those of its own `CustomRCNNRecurrent` run (tests/test_fullsize_gpu.py).  Pyramids are LEVEL MAJOR over the scenes
([level][scene][h*w] rows: the [B,h,w,256] images the N = B convs write), which is the one layout change against the single-scene
model.  `modeling/batched.py` (B scene objects on B streams, only the trunk batched) stays as the alternative schedule.

A sequence whose episode has ended (ragged episodes) stays in the batch as an idle slot: it is fed its last frame again, its memory
selection is emptied before the write (a write without instances leaves the state untouched, custom_rcnn.py:689-690) and its
results are dropped."""
from __future__ import annotations

import time as _time
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _lib, ops
from ..structures import Boxes, Instances
from .meta_arch import CustomRCNNRecurrent, _det_stream, _sched_streams

PYRAMID_SETS = 3      # the step's own, the one computed ahead, the previous step's (its detection pass may trail)
RESULT_SETS = 3


class _SceneView:
    """What tests and drivers read of one scene of the batch: its recurrent state."""

    def __init__(self, owner: "LockstepScenes", b: int):
        self._o, self._b = owner, b

    @property
    def implicit_memory(self):
        return None if self._o.implicit_memory is None else self._o.implicit_memory[self._b]

    @property
    def observations(self):
        return None if self._o.observations is None else self._o.observations[self._b]


class LockstepScenes:
    """`LockstepScenes(cfg, B)(episodes)`: `episodes` = list of B frame lists, one per sequence (`None` or `[]`: that sequence sits
    this call out; lengths may differ); returns a list of B output lists, each what `CustomRCNNRecurrent.forward([episode_b])`
    returns."""

    def __init__(self, cfg, batch: int, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        if batch < 1 or batch > _lib.MAX_BATCH:
            raise ValueError(f"batch must be in [1, {_lib.MAX_BATCH}]")
        self.B = int(batch)
        # the layers, their weights and the configuration exist once; its single-scene buffers are not used
        self.model = CustomRCNNRecurrent(cfg, state_dict)
        m = self.model
        self.device = m.device
        self.scenes = [_SceneView(self, b) for b in range(self.B)]
        self.trunk_lookahead = True          # step t + 1's memory-independent trunk on its own stream beside step t's chain
        self.trail_detection_pass = True     # step t's detection mask pass + paste under step t + 1's latency-bound front
        # True: every layer is planned like ONE scene (same split-K slabs / wave split, same summation order): each scene's results are
        # bitwise those of its own single-scene run.  False: layers are planned for the rows the batched launch really has (fewer
        # slabs and reduce launches, 64x64 tiles instead of the wave-split kernel on the B x R rows of the FC layers): the same
        # arithmetic in another summation order -- results agree with the single-scene run like two fp32 implementations do
        # (tests/test_fullsize_gpu.py::test_lockstep_planned_for_the_batch_agrees_within_tolerance)
        self.plan_like_single = True
        self.implicit_memory: Optional[torch.Tensor] = None     # [B,N,512]
        self.observations: Optional[torch.Tensor] = None        # [B,N]
        self._mem_f16 = self._dirty = None
        self._f16_valid = False
        self._dirty_pending = False
        self._err = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self._bufs = None
        self._pyramid = 0
        self._prefetched = None              # tuple of the image objects whose trunk has been computed into the next pyramid set
        self._trunk_stream = self._det_stream = None
        self._ev_trunk = torch.cuda.Event()
        self._ev_start = torch.cuda.Event()
        self._ev_box = torch.cuda.Event()
        self._ev_det = [None] * RESULT_SETS
        self._pyr_reader = {}
        self._slot = 0
        self._step_no = 0
        self.stats_log = None
        self.trace = None                    # diagnostics: list of (step, name, timing event), see tools/frame_schedule.py
        self.host_profile = {"frames": 0, "enqueue_s": 0.0, "materialize_s": 0.0, "wait_s": 0.0}
        if bool(cfg.MODEL.TEST_SAVE_SEMMAP):
            raise NotImplementedError("MODEL.TEST_SAVE_SEMMAP is served by the single-scene model (custom_rcnn.py:518-530)")
        R, C1, dev, B = m.proposal_generator.cap, m.C1, self.device, self.B
        rh = m.roi_heads
        f32 = dict(dtype=torch.float32, device=dev)
        self.R, self.D = R, rh.topk
        # cascade buffers, B x the single-scene ones
        self.pool7 = torch.empty((B * R, 7, 7, 256), **f32)
        self.h1 = torch.empty((B * R, 1, 1, 1024), **f32)
        self.h2 = torch.empty((B * R, 1, 1, 1024), **f32)
        self.hb = torch.empty((B * R, 1, 1, 1024), **f32)
        self.feat = torch.empty((B * R, 1, 1, 512), **f32)
        self.feat0 = torch.empty((B * R, 1, 1, 512), **f32)
        self.featn0 = torch.zeros((B * R, 512), **f32)
        self.deltas = torch.empty((B * R, 1, 1, 4), **f32)
        self.prob = torch.zeros((B * R, C1), **f32)
        self.boxes = [torch.zeros((B * R, 4), **f32) for _ in range(rh.num_stages + 1)]
        self.mem_scores = torch.zeros((B * R, C1), **f32)
        self.selectors = [ops.DetectionSelector(R, C1, rh.topk, dev, groups=True, batch=B) for _ in range(RESULT_SETS)]
        self.mem_selector = ops.DetectionSelector(R, C1, 100, dev, unique=True, batch=B)
        # mask passes over the concatenated lists of all scenes
        self.Pcap = min(R, 128)                       # memory instances per scene: <= 100 unique rows
        i32 = dict(dtype=torch.int32, device=dev)
        self.glist_p = torch.zeros((B * R,), **i32)
        self.total_p = torch.zeros((1,), **i32)
        self.glist_d = [torch.zeros((B * rh.topk,), **i32) for _ in range(RESULT_SETS)]
        self.total_d = [torch.zeros((1,), **i32) for _ in range(RESULT_SETS)]
        self.pm_bufs = (torch.empty((B * self.Pcap, 14, 14, 256), **f32), torch.empty((B * self.Pcap, 14, 14, 256), **f32))
        self.dm_bufs = (torch.empty((B * rh.topk, 14, 14, 256), **f32), torch.empty((B * rh.topk, 14, 14, 256), **f32))
        self.prop_masks = torch.zeros((B * R, 28, 28), **f32)
        self.det_masks = torch.zeros((B * rh.topk, 28, 28), **f32)

    # nn.Module surface of the drivers
    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __call__(self, episodes):
        return self.forward(episodes)

    def _pr(self, rows: int) -> int:
        return rows if self.plan_like_single else 0

    def _mark(self, name: str, stream=None):
        if self.trace is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream(self.device))
        self.trace.append((self._step_no, name, ev))

    # ---- buffers that depend on the frame size -----------------------------------------------------------------------------
    def _frame_buffers(self, H: int, W: int, n_cells: int):
        key = (H, W, n_cells)
        if self._bufs is not None and self._bufs["key"] == key:
            return self._bufs
        m, B, dev = self.model, self.B, self.device
        bb, pg = m.backbone, m.proposal_generator
        shapes = bb.level_shapes(H, W)
        off = [0]
        for (h, w) in shapes:
            off.append(off[-1] + h * w)
        P = off[-1]
        # level-major row list of the batch: level l holds the B images [B,h,w,256] at rows [B*off[l], B*off[l+1])
        offB, shapesB = [0], []
        for l, (h, w) in enumerate(shapes):
            for _b in range(B):
                offB.append(offB[-1] + h * w)
                shapesB.append((h, w))
        pyr = []
        for _ in range(PYRAMID_SETS):
            feats = torch.empty((B * P, 256), dtype=torch.float32, device=dev)
            views = [feats[B * off[i]:B * off[i + 1]].view(B, shapes[i][0], shapes[i][1], 256) for i in range(5)]
            pyr.append((feats, views))
        d = dict(key=key, shapes=shapes, off=off, P=P, offB=offB, shapesB=shapesB, pyr=pyr)
        d["x"] = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
        d["proj"] = torch.zeros((B, H, W), dtype=torch.int32, device=dev)
        d["pooled"] = torch.empty((B * ops.pooled_rows(H, W), 512), dtype=torch.float16, device=dev)
        d["tower_a"] = torch.empty((B * P, 256), dtype=torch.float32, device=dev)
        d["tower_b"] = torch.empty((B * P, 256), dtype=torch.float32, device=dev)
        d["head"] = torch.empty((B * P, 5), dtype=torch.float32, device=dev)
        d["gn_ws"] = ops.groupnorm_workspace(offB, dev)
        d["dec"] = ops.ProposalDecoder(shapes, pg.strides, pg.scales, pg.score_thresh, pg.pre_nms_topk, pg.post_nms_topk, pg.nms_thresh,
                                       pg.cap, dev, head_stride=5, batch=B)
        d["writer"] = ops.MemoryWriter(H, W, n_cells, 100, self.R, dev, mask_thresh=0.5, batch=B)
        D = self.D
        d["posts"] = [dict(hw=(H, W), boxes=None, scores=None, classes=None, masks=None,
                           src=torch.zeros((B * D,), dtype=torch.int32, device=dev), count=torch.zeros((B,), dtype=torch.int32, device=dev),
                           count_host=torch.zeros((B,), dtype=torch.int32).pin_memory(),
                           err_host=torch.zeros((1,), dtype=torch.int32).pin_memory(), ready=torch.cuda.Event(),
                           err_ready=torch.cuda.Event()) for _ in range(RESULT_SETS)]
        self._bufs = d
        self._prefetched = None
        return d

    # ---- state -------------------------------------------------------------------------------------------------------------
    def _ensure_state(self, n_cells: int):
        if self.implicit_memory is None or self.implicit_memory.shape[1] != n_cells:
            B, dev = self.B, self.device
            self.implicit_memory = torch.zeros((B, n_cells, 512), dtype=torch.float32, device=dev)
            self.observations = torch.zeros((B, n_cells), dtype=torch.float32, device=dev)
            self._mem_f16 = torch.zeros((B, n_cells, 512), dtype=torch.float16, device=dev)
            self._dirty = torch.zeros((B, n_cells), dtype=torch.int32, device=dev)
            self._f16_valid = True
            self._dirty_pending = False
            self._started = [False] * B

    def reset_memory(self, b: int):
        """`frame['memory_reset']` branch (custom_rcnn.py:470-479) for scene b."""
        lib, s = _lib.load(), torch.cuda.current_stream(self.device).cuda_stream
        mem, obs, m16, dirty = self.implicit_memory[b], self.observations[b], self._mem_f16[b], self._dirty[b]
        _lib.check(lib.eod_fill_f32(mem.data_ptr(), 0.0, mem.numel(), s), "fill")
        _lib.check(lib.eod_fill_f32(obs.data_ptr(), 0.0, obs.numel(), s), "fill")
        _lib.check(lib.eod_fill_i32(m16.data_ptr(), 0, m16.numel() // 2, s), "fill")
        _lib.check(lib.eod_fill_i32(dirty.data_ptr(), 0, dirty.numel(), s), "fill")
        self._started[b] = True

    def invalidate_memory_snapshot(self):
        self._f16_valid = False

    def _refresh_snapshot(self):
        """a4 + fp16 cast for the B tables: they are one [B*N,512] table to the kernels."""
        mem, obs = self.implicit_memory.view(-1, 512), self.observations.view(-1)
        m16, dirty = self._mem_f16.view(-1, 512), self._dirty.view(-1)
        if self._f16_valid:
            if self._dirty_pending:
                ops.memory_normalize_dirty_f16(mem, obs, dirty, m16)
        else:
            ops.memory_normalize_f16(mem, obs, out=m16)
            _lib.check(_lib.load().eod_fill_i32(dirty.data_ptr(), 0, dirty.numel(), torch.cuda.current_stream().cuda_stream), "fill")
            self._f16_valid = True
        self._dirty_pending = False

    # ---- the memory-independent half -----------------------------------------------------------------------------------------
    def _trunk(self, frames: List[dict], H: int, W: int, which: int):
        """preprocess + ResNet-50 + FPN top-down for the B images, N = B, into pyramid set `which` (level-major views)."""
        m, B, d = self.model, self.B, self._bufs
        bb = m.backbone
        x = d["x"]
        for b, f in enumerate(frames):
            ops.preprocess_image(m._device_image(f), m.pixel_mean, m.pixel_std, out=x[b:b + 1])
        c = bb.bottom_up.forward(x, H, W, N=B, plan_like_single=self.plan_like_single)
        (c5, h5, w5), (c4, h4, w4), (c3, h3, w3) = c["layer5"], c["layer4"], c["layer3"]
        views = d["pyr"][which][1]
        lat5 = bb.lateral[5](c5, B, h5, w5, plan_rows=self._pr(h5 * w5))
        bb.output[5](lat5, B, h5, w5, out=views[2], plan_rows=self._pr(h5 * w5))
        lat4 = bb.lateral[4](c4, B, h4, w4, res=lat5, res_mode=2, plan_rows=self._pr(h4 * w4))
        bb.output[4](lat4, B, h4, w4, out=views[1], plan_rows=self._pr(h4 * w4))
        lat3 = bb.lateral[3](c3, B, h3, w3, res=lat4, res_mode=2, plan_rows=self._pr(h3 * w3))
        bb.output[3](lat3, B, h3, w3, out=views[0], plan_rows=self._pr(h3 * w3))

    def _enqueue_trunk_ahead(self, frames: List[dict], H: int, W: int):
        if self._trunk_stream is None:
            self._trunk_stream = _sched_streams(self.device)[1]
        ts = self._trunk_stream
        nxt = (self._pyramid + 1) % PYRAMID_SETS
        ts.wait_event(self._ev_start)
        if nxt in self._pyr_reader:
            ts.wait_event(self._pyr_reader[nxt])            # a trailing detection pass may still read that set
        with torch.cuda.stream(ts):
            self._mark("trunk_lookahead_begin", ts)
            self._trunk(frames, H, W, nxt)
            self._ev_trunk.record(ts)
            self._mark("trunk_lookahead", ts)
        self._prefetched = tuple(id(f["image"]) for f in frames)

    # ---- one step: one frame of every scene ----------------------------------------------------------------------------------
    def _step(self, frames: List[dict], active: List[bool], refresh: bool, next_frames: Optional[List[dict]], trailing: bool):
        m, B, dev = self.model, self.B, self.device
        bb, pg, rh = m.backbone, m.proposal_generator, m.roi_heads
        H, W = int(frames[0]["image"].shape[-2]), int(frames[0]["image"].shape[-1])
        if H % 32 or W % 32:
            raise ValueError("H and W must be multiples of 32 (proj_indices is not padded: SURVEY §8 notation)")
        if any((int(f["image"].shape[-2]), int(f["image"].shape[-1])) != (H, W) for f in frames):
            raise ValueError("the frames of one lock-step must have one size")
        n_cells = self.implicit_memory.shape[1]
        d = self._frame_buffers(H, W, n_cells)
        cur = torch.cuda.current_stream(dev)
        R, D, C1 = self.R, self.D, m.C1
        self._step_no += 1
        self._mark("start")
        for b, f in enumerate(frames):
            p = m._device_proj(f)
            if tuple(p.shape) != (H, W):
                raise ValueError(f"proj_indices shape {tuple(p.shape)} != image {(H, W)}")
            d["proj"][b].copy_(p, non_blocking=True)
        use_mem = m.memory_type == "implicit_memory"
        if use_mem and refresh:
            self._refresh_snapshot()

        key = tuple(id(f["image"]) for f in frames)
        had_ahead = self._prefetched is not None
        hit = had_ahead and self._prefetched == key
        self._prefetched = None
        if had_ahead:
            cur.wait_event(self._ev_trunk)                      # used or not, it must be over before any set is touched
        self._pyramid = (self._pyramid + 1) % PYRAMID_SETS
        which = self._pyramid
        if not hit:
            if which in self._pyr_reader:
                cur.wait_event(self._pyr_reader[which])
            self._trunk(frames, H, W, which)
        feats, views = d["pyr"][which]
        shapes, off, P = d["shapes"], d["off"], d["P"]
        look = next_frames is not None and self.trunk_lookahead
        if look:
            self._ev_start.record(cur)

        # memory read + fusion (timm.py:142-192), P6 / P7 (timm.py:359-364)
        if use_mem and bb.feat_fusion != "image_only":
            ops.memory_gather_pool(self._mem_f16, d["proj"], H, W, out=d["pooled"], err=self._err, torch_order=bb.pool_in_torch_order,
                                   batch=B)
            bb.merge(d["pooled"], feats, H, W, bb.map_feature_weight, bb.feat_fusion, batch=B)
        (h5, w5), (h6, w6), (h7, w7) = shapes[2], shapes[3], shapes[4]
        bb.p6(views[2], B, h5, w5, out=views[3], plan_rows=self._pr(h6 * w6))
        bb.p7(views[3], B, h6, w6, in_relu=True, out=views[4], plan_rows=self._pr(h7 * w7))

        # CenterNet tower + proposals (centernet_head.py:141-161, centernet.py:603-745)
        lv = (d["offB"], d["shapesB"])
        src = feats
        for (conv, gamma, beta) in pg.tower:
            conv(src, 1, 0, 0, out=d["tower_a"], levels=lv, plan_rows=self._pr(P), gn_stats=d["gn_ws"] if pg.fuse_gn_stats else None)
            ops.groupnorm_relu(d["tower_a"], gamma, beta, d["offB"], 256, d["gn_ws"], out=d["tower_b"], partial_ready=conv.gn_fused)
            src = d["tower_b"]
        pg.out_conv(src, 1, 0, 0, out=d["head"], levels=lv, plan_rows=self._pr(P))
        prop_boxes, prop_scores, prop_count = d["dec"](d["head"])
        self._mark("proposals")
        if look:
            self._enqueue_trunk_ahead(next_frames, H, W)

        # cascade (detic_roi_heads.py:88-222)
        h3, w3 = shapes[0]
        k = self._slot
        if self._ev_det[k] is not None:
            cur.wait_event(self._ev_det[k])                     # detection list set k is still read by the pass of RESULT_SETS steps ago
        update_mem = use_mem or m.always_update_memory
        boxes = prop_boxes
        seg = dict(m_count=prop_count, m_unit=1, m_segments=B, plan_rows=self._pr(R))
        pending = None
        for s_i, st in enumerate(rh.stages):
            if pending is None:
                ops.roi_align(views[0], views[1], views[2], h3, w3, 256, boxes, prop_count, B * R, 7, out=self.pool7, batch=B,
                              boxes_per_image=R)
            else:       # the previous stage's deltas are applied by the ROIAlign launch itself (roi_heads.fold_deltas)
                ops.roi_align(views[0], views[1], views[2], h3, w3, 256, boxes, prop_count, B * R, 7, out=self.pool7, batch=B,
                              boxes_per_image=R, refine=(self.deltas, 4, pending, True, float(W), float(H), self.boxes[s_i]))
                boxes = self.boxes[s_i]
                pending = None
            st["fc1"](self.pool7, B * R, 1, 1, relu=True, out=self.h1, **seg)
            st["fc2"](self.h1, B * R, 1, 1, relu=True, out=self.h2, **seg)
            feat = self.feat0 if s_i == 0 else self.feat
            st["cls_bb0"](self.h2, B * R, 1, 1, relu=True, out=feat, split=(512, self.hb), **seg)
            last = s_i == rh.num_stages - 1
            rescore = s_i == 0 and update_mem
            if rh.fuse_stage_tail and not rh.fold_deltas:
                # classifier tail + bbox_pred.2 + apply_deltas in one launch, as the single-scene model runs them (roi_heads._cascade)
                ops.cascade_stage_tail(feat, st["zs"], self.prob, s_i > 0, self.featn0 if s_i == 0 else None, prop_count, R, C1, rh.norm_temp,
                                       self.hb, st["bb2"], boxes, self.boxes[s_i + 1], rh.cascade_weights[s_i], not last, float(W), float(H),
                                       zs_mem=m.zs_weight if rescore else None, prop_scores=prop_scores if (last or rescore) else None,
                                       mem_scores_out=self.mem_scores if rescore else None,
                                       final_inv_stages=1.0 / rh.num_stages if last else 0.0, deltas_out=self.deltas, batch=B)
                boxes = self.boxes[s_i + 1]
                continue
            ops.zs_classify(feat, st["zs"], self.prob, s_i > 0, self.featn0 if s_i == 0 else None, prop_count, R, C1, rh.norm_temp,
                            zs_mem=m.zs_weight if rescore else None, prop_scores=prop_scores if (last or rescore) else None,
                            mem_scores_out=self.mem_scores if rescore else None,
                            final_inv_stages=1.0 / rh.num_stages if last else 0.0, batch=B)
            st["bb2"](self.hb, B * R, 1, 1, out=self.deltas, **seg)
            if rh.fold_deltas and not last:
                pending = rh.cascade_weights[s_i]
            else:
                ops.apply_deltas(self.deltas, 4, boxes, self.boxes[s_i + 1], prop_count, R, rh.cascade_weights[s_i], not last, float(W),
                                 float(H), batch=B)
                boxes = self.boxes[s_i + 1]
        # memory selection first: the step's critical chain waits for it (custom_rcnn.py:825-875)
        msel = self.mem_selector
        if update_mem:
            _, _, _, mem_rows, mem_cnt = msel(prop_boxes, self.mem_scores, prop_count, float(W), float(H), m.cls_score_thresh, 0.5)
            for b in range(B):
                if not active[b]:                               # idle slot: no instances -> its state stays as it is
                    s_raw = torch.cuda.current_stream(dev).cuda_stream
                    _lib.check(_lib.load().eod_fill_i32(mem_cnt[b:b + 1].data_ptr(), 0, 1, s_raw), "fill")
                    _lib.check(_lib.load().eod_fill_i32(msel.uniq_count[b:b + 1].data_ptr(), 0, 1, s_raw), "fill")
        self._mark("cascade+mem_select")
        sel = self.selectors[k]
        det_boxes, det_scores, det_classes, det_rows, det_count = sel(boxes, self.prob, prop_count, float(W), float(H), rh.score_thresh,
                                                                     rh.nms_thresh)
        self._ev_box.record(cur)

        # detection mask pass + post-processing + paste (detic_roi_heads.py:257, custom_rcnn.py:579-580)
        post = d["posts"][k]
        if self.trail_detection_pass:
            if self._det_stream is None:
                self._det_stream = _det_stream(dev, m.det_stream_priority)
                self._ev_det = [torch.cuda.Event() for _ in range(RESULT_SETS)]
            ds = self._det_stream
            ds.wait_event(self._ev_box)
            with torch.cuda.stream(ds):
                self._mark("det_pass_begin", ds)
                self._detection_pass(views, h3, w3, sel, k, H, W, post)
                self._ev_det[k].record(ds)
                self._mark("det_pass", ds)
            self._pyr_reader[which] = self._ev_det[k]
        else:
            self._detection_pass(views, h3, w3, sel, k, H, W, post)

        # mask head on the memory instances of all scenes (custom_rcnn.py:573-574, only the proposals 875-880 read), memory write
        if update_mem:
            ops.concat_lists(msel.uniq_rows, msel.uniq_count, R, R, B, self.glist_p, self.total_p)
            self._mask_pass(views, h3, w3, prop_boxes, self.glist_p, self.total_p, B * self.Pcap, R, self.prop_masks, self.pm_bufs,
                            plan_rois=self.Pcap, tag=("prop", self._step_no))
            self._mark("prop_masks")
            follow = m.snapshot_follows_write if m.snapshot_follows_write is not None else m.test_type in ("default", "episodic")
            wr = d["writer"]
            if follow and self._f16_valid and not self._dirty_pending:
                wr(self.featn0, prop_boxes, self.prop_masks, mem_rows, mem_cnt, d["proj"], self.implicit_memory, self.observations,
                   err=self._err, snapshot=self._mem_f16)
            else:
                wr(self.featn0, prop_boxes, self.prop_masks, mem_rows, mem_cnt, d["proj"], self.implicit_memory, self.observations,
                   dirty=self._dirty, err=self._err)
                self._dirty_pending = True
            self._mark("mem_write")
        if self.trail_detection_pass and not trailing:
            cur.wait_event(self._ev_det[k])
        if self.stats_log is not None:
            if self.trail_detection_pass:
                with torch.cuda.stream(self._det_stream):
                    cnt = post["count"].clone()
            else:
                cnt = post["count"].clone()
            self.stats_log.append((prop_count.clone(), cnt, d["writer"].k_out.clone(), msel.uniq_count.clone(), sel.rep_count.clone()))
        return self._ticket(post, k)

    def _mask_pass(self, views, h3, w3, boxes, glist, total, cap, boxes_per_image, out, bufs, plan_rois, tag=None):
        """The mask head (4 convs + deconv + predictor + sigmoid) once over the concatenated ROI lists of all scenes: ROI i pools box
        glist[i] (a global index b * boxes_per_image + row) from image b and its 28x28 probabilities go to out[glist[i]]."""
        rh = self.model.roi_heads
        B = self.B
        src, dst = bufs
        ops.roi_align(views[0], views[1], views[2], h3, w3, 256, boxes, total, cap, 14, out=src, box_rows=glist, batch=B,
                      boxes_per_image=boxes_per_image)
        for conv in rh.mask_convs:
            conv.event_tag = tag
            conv(src, cap, 14, 14, relu=True, m_count=total, m_unit=196, out=dst, plan_rows=self._pr(plan_rois * 196))
            src, dst = dst, src
        rh.deconv(src, cap, 14, 14, relu=True, m_count=total, m_unit=196, out=out, fuse=(rh.pred_w, rh.pred_b, glist),
                  plan_rows=self._pr(plan_rois * 196))

    def _detection_pass(self, views, h3, w3, sel, k, H, W, post):
        m, B, D, dev = self.model, self.B, self.D, self.device
        ops.concat_lists(sel.rep_list, sel.rep_count, D, D, B, self.glist_d[k], self.total_d[k])
        self._mask_pass(views, h3, w3, sel.boxes, self.glist_d[k], self.total_d[k], B * D, D, self.det_masks, self.dm_bufs, plan_rois=D,
                        tag=("det", self._step_no))
        post["boxes"] = torch.empty((B * D, 4), dtype=torch.float32, device=dev)
        post["scores"] = torch.empty((B * D,), dtype=torch.float32, device=dev)
        post["classes"] = torch.empty((B * D,), dtype=torch.int32, device=dev)
        post["masks"] = torch.empty((B, D, H, W), dtype=torch.uint8, device=dev)
        ops.detector_postprocess(sel.boxes, sel.scores, sel.classes, sel.count, D, 1.0, 1.0, float(W), float(H), post["boxes"],
                                 post["scores"], post["classes"], post["src"], post["count"], remap=sel.rep_of, batch=B)
        ops.paste_masks(self.det_masks, post["boxes"], post["src"], post["count"], D, H, W, m.mask_threshold, post["masks"], batch=B,
                        prob_units=D)

    def _ticket(self, post, k):
        """Async read-back of the step's detection counts into pinned host memory + an event; flips the result set."""
        cur = torch.cuda.current_stream(self.device)
        post["err_host"].copy_(self._err, non_blocking=True)
        if self.trail_detection_pass:
            ds = self._det_stream
            post["err_ready"].record(cur)
            ds.wait_event(post["err_ready"])
            with torch.cuda.stream(ds):
                post["count_host"].copy_(post["count"], non_blocking=True)
                post["ready"].record(ds)
        else:
            post["count_host"].copy_(post["count"], non_blocking=True)
            post["ready"].record(cur)
        self._slot = (self._slot + 1) % RESULT_SETS
        return post

    def _materialize(self, post, active: List[bool]) -> List[Optional[Instances]]:
        t0 = _time.perf_counter()
        post["ready"].synchronize()
        self.host_profile["wait_s"] += _time.perf_counter() - t0
        flags = int(post["err_host"][0])
        if flags:
            self._err.zero_()
            raise _lib.EodError(
                f"device error flags {flags:#x}: proj_indices holds cell indices outside [0, {self.implicit_memory.shape[1]}) "
                "(an index image written for another map size?); they were clamped, the step's results are not trustworthy")
        cur = torch.cuda.current_stream(self.device)
        for key in ("boxes", "scores", "classes", "masks"):
            post[key].record_stream(cur)
        D = self.D
        out = []
        for b in range(self.B):
            if not active[b]:
                out.append(None)
                continue
            n = int(post["count_host"][b])
            inst = Instances(post["hw"])
            inst.pred_boxes = Boxes(post["boxes"][b * D:b * D + n])
            inst.scores = post["scores"][b * D:b * D + n]
            inst.pred_classes = post["classes"][b * D:b * D + n].to(torch.int64)
            inst.pred_masks = post["masks"][b, :n].view(torch.bool)
            out.append(inst)
        post["boxes"] = post["scores"] = post["classes"] = post["masks"] = None
        return out

    # ---- forward ---------------------------------------------------------------------------------------------------------------
    def forward(self, episodes: List[Optional[List[dict]]]):
        B, m = self.B, self.model
        if len(episodes) != B:
            raise ValueError(f"need {B} episodes (None for a sequence that sits this call out)")
        episodes = [e if e else [] for e in episodes]
        T = max(len(e) for e in episodes)
        outs: List[List[dict]] = [[] for _ in range(B)]
        if T == 0:
            return outs
        first = next(e[0] for e in episodes if e)
        self._ensure_state(int(first["memory"].shape[0]))
        # the step's chain runs on the process-wide high-priority chain stream; the caller's stream is joined on both sides
        caller = torch.cuda.current_stream(self.device)
        ms = _sched_streams(self.device)[2]
        ev_in, ev_out = torch.cuda.Event(), torch.cuda.Event()
        ev_in.record(caller)
        ms.wait_event(ev_in)
        idle = dict(image=first["image"], proj_indices=first["proj_indices"])

        def frames_at(t):
            fr, act = [], []
            for b in range(B):
                e = episodes[b]
                act.append(t < len(e))
                fr.append(e[t] if t < len(e) else (e[-1] if e else idle))
            return fr, act

        pending = []
        with torch.cuda.stream(ms):
            for t in range(T):
                frames, active = frames_at(t)
                t0 = _time.perf_counter()
                for b in range(B):
                    if active[b] and frames[b]["memory_reset"]:
                        if int(episodes[b][0]["memory"].shape[0]) != self.implicit_memory.shape[1]:
                            raise ValueError("the scenes of one lock-step must have one memory size")
                        self.reset_memory(b)
                    if active[b] and not self._started[b]:
                        raise RuntimeError("first frame of a scene must carry memory_reset=True (custom_rcnn.py:485 reads unset state)")
                refresh = m.test_type in ("default", "episodic") or (m.test_type == "longterm" and t == 0)
                nxt = frames_at(t + 1)[0] if t + 1 < T else None
                pending.append((self._step(frames, active, refresh, nxt, trailing=t + 1 < T), active))
                t1 = _time.perf_counter()
                if len(pending) == RESULT_SETS:
                    post, act = pending.pop(0)
                    for b, inst in enumerate(self._materialize(post, act)):
                        if inst is not None:
                            outs[b].append({"instances": inst})
                hp = self.host_profile
                hp["frames"] += sum(active)
                hp["enqueue_s"] += t1 - t0
                hp["materialize_s"] += _time.perf_counter() - t1
            for post, act in pending:
                for b, inst in enumerate(self._materialize(post, act)):
                    if inst is not None:
                        outs[b].append({"instances": inst})
            ev_out.record(ms)
        caller.wait_event(ev_out)
        return outs
