"""DeticCascadeROIHeads (inference) on the HIP kernels.

Mirrors `_forward_box/_run_stage/_create_proposals_from_boxes/forward/forward_mask_memory`
(`Detic/detic/modeling/roi_heads/detic_roi_heads.py:88-349`), detectron2's `ROIPooler`(ROIAlignV2),
`FastRCNNConvFCHead`, `MaskRCNNConvUpsampleHead`, `Box2BoxTransform`, `fast_rcnn_inference`,
`DeticFastRCNNOutputLayers.forward/predict_probs` (`detic_fast_rcnn.py:437-466,325-339`) and `ZeroShotClassifier.forward`
(`zero_shot_classifier.py:71-111`).  ROI lists have a fixed capacity and a device-side count; there is no host sync
anywhere in here.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .. import ops
from ..registry import ROI_HEADS_REGISTRY


@ROI_HEADS_REGISTRY.register()
class DeticCascadeROIHeads:
    def __init__(self, cfg, sd: Dict[str, torch.Tensor], device, prop_cap: int):
        self.device = device
        rb = cfg.MODEL.ROI_BOX_HEAD
        if not (rb.USE_ZEROSHOT_CLS and rb.CLS_AGNOSTIC_BBOX_REG and rb.USE_SIGMOID_CE and rb.MULT_PROPOSAL_SCORE):
            raise NotImplementedError("hot path covers the zero-shot, class-agnostic, sigmoid, mult-proposal-score cascade")
        if rb.NUM_FC != 2 or rb.POOLER_RESOLUTION != 7 or cfg.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION != 14:
            raise NotImplementedError("unsupported ROI head geometry")
        if not cfg.MODEL.ROI_MASK_HEAD.CLS_AGNOSTIC_MASK or cfg.MODEL.ROI_MASK_HEAD.NUM_CONV != 4:
            raise NotImplementedError("unsupported mask head")
        self.num_classes = int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        self.C1 = self.num_classes + 1
        self.norm_temp = float(rb.NORM_TEMP)
        self.cascade_weights = [tuple(float(v) for v in w) for w in cfg.MODEL.ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS]
        self.num_stages = len(self.cascade_weights)
        self.score_thresh = float(cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST)
        self.nms_thresh = float(cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST)
        self.topk = int(cfg.TEST.DETECTIONS_PER_IMAGE)
        self.add_feature_to_prop = bool(rb.ADD_FEATURE_TO_PROP)
        self.R = prop_cap                       # proposal capacity
        self.Dcap = (self.topk + 31) // 32 * 32  # detection capacity for the mask head
        self.stages = []
        for k in range(self.num_stages):
            fc1_w = sd[f"roi_heads.box_head.{k}.fc1.weight"]
            # flatten order: reference (C,7,7) -> ours (7,7,C)
            fc1_w = fc1_w.view(-1, 256, 7, 7).permute(0, 2, 3, 1).reshape(fc1_w.shape[0], -1, 1, 1).contiguous()
            p = f"roi_heads.box_predictor.{k}"
            st = dict(
                fc1=ops.Conv(fc1_w, sd[f"roi_heads.box_head.{k}.fc1.bias"], device=device, name=f"box_head.{k}.fc1"),
                fc2=ops.Conv(sd[f"roi_heads.box_head.{k}.fc2.weight"][:, :, None, None], sd[f"roi_heads.box_head.{k}.fc2.bias"],
                             device=device, name=f"box_head.{k}.fc2"),
                cls=ops.Conv(sd[f"{p}.cls_score.linear.weight"][:, :, None, None], sd[f"{p}.cls_score.linear.bias"], device=device,
                             name=f"box_predictor.{k}.cls_score.linear"),
                bb0=ops.Conv(sd[f"{p}.bbox_pred.0.weight"][:, :, None, None], sd[f"{p}.bbox_pred.0.bias"], device=device,
                             name=f"box_predictor.{k}.bbox_pred.0"),
                bb2=ops.Conv(sd[f"{p}.bbox_pred.2.weight"][:, :, None, None], sd[f"{p}.bbox_pred.2.bias"], device=device,
                             name=f"box_predictor.{k}.bbox_pred.2"),
                zs=sd[f"{p}.cls_score.zs_weight"].contiguous().to(device),
            )
            # cls_score.linear (1024 -> 512, no ReLU) and bbox_pred.0 (1024 -> 1024, ReLU) read the same fc2 output: one GEMM with
            # the weights stacked, two outputs (EodConvDesc.split_n); each output column walks K as in the separate call
            st["cls_bb0"] = ops.Conv(torch.cat([sd[f"{p}.cls_score.linear.weight"], sd[f"{p}.bbox_pred.0.weight"]], dim=0)[:, :, None, None],
                                     torch.cat([sd[f"{p}.cls_score.linear.bias"], sd[f"{p}.bbox_pred.0.bias"]], dim=0), device=device,
                                     name=f"box_predictor.{k}.cls_score.linear+bbox_pred.0")
            assert st["zs"].shape[1] == self.C1, (st["zs"].shape, self.C1)
            self.stages.append(st)
        m = "roi_heads.mask_head"
        self.mask_convs = [ops.Conv(sd[f"{m}.mask_fcn{i}.weight"], sd[f"{m}.mask_fcn{i}.bias"], pad=1, device=device, name=f"mask_fcn{i}")
                           for i in range(1, 5)]
        # EodConvDesc.prefetch2 (two chunks of operands in flight) was built for these launches (~40 / ~90 ROIs leave 2-5 workgroups
        # on a CU) and measured slower at every size (tools/conv_bench.py propmask: 43 ROIs 128 against 121 us, 100 ROIs 254
        # against 229, 300 ROIs 688 against 644; whole frame 273 against 281 frames/s): off; EOD_MASK_PREFETCH2=1 turns it on
        for conv in self.mask_convs:
            conv.prefetch2 = int(__import__("os").environ.get("EOD_MASK_PREFETCH2", "0"))
        self.deconv = ops.Conv(sd[f"{m}.deconv.weight"], sd[f"{m}.deconv.bias"], device=device, deconv=True, name="mask_deconv")
        self.pred_w = sd[f"{m}.predictor.weight"].reshape(-1).contiguous().to(device)
        self.pred_b = float(sd[f"{m}.predictor.bias"].item())
        # persistent buffers
        R, D = self.R, self.Dcap
        f32 = dict(dtype=torch.float32, device=device)
        self.pool7 = torch.empty((R, 7, 7, 256), **f32)
        self.h1 = torch.empty((R, 1, 1, 1024), **f32)
        self.h2 = torch.empty((R, 1, 1, 1024), **f32)
        self.hb = torch.empty((R, 1, 1, 1024), **f32)
        self.feat = torch.empty((R, 1, 1, 512), **f32)
        self.feat0 = torch.empty((R, 1, 1, 512), **f32)        # stage-0 CLIP-space feature = proposals.feat
        self.featn0 = torch.zeros((R, 512), **f32)            # 50 * normalize(feat0): what the memory stores
        self.deltas = torch.empty((R, 1, 1, 4), **f32)
        self.prob = torch.zeros((R, self.C1), **f32)
        self.boxes = [torch.zeros((R, 4), **f32) for _ in range(self.num_stages + 1)]
        M = max(R, D)
        self.mpool = torch.empty((M, 14, 14, 256), **f32)
        self.mbuf = torch.empty((M, 14, 14, 256), **f32)
        self.mup = torch.empty((M, 28, 28, 256), **f32)
        self.det_masks = torch.zeros((D, 28, 28), **f32)
        self.prop_masks = torch.zeros((R, 28, 28), **f32)
        self._prop_bufs = None      # second activation set, allocated when the proposal pass runs on its own stream
        # deconv + ReLU + predictor + sigmoid in ONE launch (the [rois,28,28,256] activation never goes to memory); False keeps
        # the two-launch form (used by the tests as the cross-check)
        self.fuse_mask_tail = True
        self.merge_cls_bb0 = True     # False: the two linear layers as two launches (tests: bitwise the same results)
        # classifier tail + bbox_pred.2 + apply_deltas of a stage in one launch (`eod_cascade_stage_tail`); False: three launches
        # (bbox_pred.2 then on the matrix cores: the deltas agree to fp32 summation order, ~1e-7 relative)
        self.fuse_stage_tail = True
        # True: the next stage's ROIAlign applies the deltas on load (EodBoxRefine: one launch less per stage, bitwise the same
        # results).  Measured in the frame (tools/knob_ab.py, same call): 287.1 frames/s against 288.6 with apply_deltas as its own
        # 5 us launch -- every one of the ROIAlign's 12 544 waves redoes the box arithmetic behind a dependent load: off.
        self.fold_deltas = False
        # experiment: the cascade's 21 launches as a captured hipGraph, replayed (see forward_box).  Measured in the frame
        # (tools/knob_ab.py, same call): 286.0 frames/s against 286.1 with stream launches -- the launch boundaries on the GPU are
        # drain + dispatch beside the other streams' kernels, which a graph does not remove, and the host is not the limit: off.
        self.graph_cascade = False
        self._graphs = {}
        # three detection-list sets: the detection mask pass of frame t may still read set t % 3 while the cascades of the next
        # frames write the others (meta_arch.py, pipeline_detection_pass / RESULT_SETS)
        # LDS reserve of the DETECTION mask pass's launches (the pass that trails under the frame's / the next frame's latency-bound
        # chains): see EodConvDesc.lds_reserve.  0 = the kernel's natural occupancy.
        self.det_pass_lds_reserve = int(__import__("os").environ.get("EOD_DET_LDS_RESERVE", "0"))
        self.prop_pass_lds_reserve = int(__import__("os").environ.get("EOD_PROP_LDS_RESERVE", "0"))
        self.selectors = [ops.DetectionSelector(R, self.C1, self.topk, device, groups=True) for _ in range(3)]
        self.selector = self.selectors[0]

    # ---- cascade box heads ------------------------------------------------------------------------
    def forward_box(self, views: List[torch.Tensor], shapes, prop_boxes: torch.Tensor, prop_scores: torch.Tensor, count: torch.Tensor,
                    image_hw: Tuple[int, int], sel: int = 0, stage0_event=None, mem_rescore=None, after_cascade=None):
        """`stage0_event` (optional torch.cuda.Event): recorded once stage 0 has produced `feat0` / `featn0` -- all that the memory
        selection (custom_rcnn.py:825-875) needs from the cascade.  `mem_rescore = (zs_weight of the meta-architecture, out [R, C1])`:
        stage 0's classifier launch also writes the memory update's CLIP re-score of the proposals (custom_rcnn.py:838-861).
        `after_cascade` (optional callable): enqueued between the cascade and the detection selection (the frame's critical chain
        waits for the memory selection, not for the detections)."""
        h3, w3 = shapes[0]
        H, W = image_hw
        R = self.R
        if self.graph_cascade and stage0_event is None:
            # Experiment (`graph_cascade`): the cascade's 21 launches as ONE hipGraph per (pyramid set, inputs) combination, captured on
            # first use and replayed.  Every buffer of the segment is static; the events around it stay outside the graph.
            key = (views[0].data_ptr(), prop_boxes.data_ptr(), prop_scores.data_ptr(), count.data_ptr(), mem_rescore is not None, H, W)
            g = self._graphs.get(key)
            if g is None:
                self._cascade(views, shapes, prop_boxes, prop_scores, count, image_hw, None, mem_rescore)     # eager once: workspaces exist
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._cascade(views, shapes, prop_boxes, prop_scores, count, image_hw, None, mem_rescore)
                self._graphs[key] = g
            g.replay()
            boxes = self.boxes[self.num_stages]
        else:
            boxes = self._cascade(views, shapes, prop_boxes, prop_scores, count, image_hw, stage0_event, mem_rescore)
        if after_cascade is not None:
            after_cascade()
        self.last_selector = self.selectors[sel]
        return self.selectors[sel](boxes, self.prob, count, float(W), float(H), self.score_thresh, self.nms_thresh)

    def _cascade(self, views, shapes, prop_boxes, prop_scores, count, image_hw, stage0_event, mem_rescore):
        """The three cascade stages (detic_roi_heads.py:88-175) -> the final boxes buffer; scores land in `self.prob`."""
        h3, w3 = shapes[0]
        H, W = image_hw
        R = self.R
        boxes = prop_boxes
        pending = None      # `fold_deltas`: regression weights of the deltas that the next ROIAlign applies to `boxes` on load
        for k, st in enumerate(self.stages):
            if pending is None:
                ops.roi_align(views[0], views[1], views[2], h3, w3, 256, boxes, count, R, 7, out=self.pool7)
            else:
                # next-stage proposals = apply_deltas(boxes, deltas) clipped to the image (detic_roi_heads.py:314): computed by the
                # ROIAlign launch itself (EodBoxRefine) and stored to boxes[k] -- one launch less per stage on the frame's chain
                ops.roi_align(views[0], views[1], views[2], h3, w3, 256, boxes, count, R, 7, out=self.pool7,
                              refine=(self.deltas, 4, pending, True, float(W), float(H), self.boxes[k]))
                boxes = self.boxes[k]
                pending = None
            st["fc1"](self.pool7, R, 1, 1, relu=True, m_count=count, m_unit=1, out=self.h1)
            st["fc2"](self.h1, R, 1, 1, relu=True, m_count=count, m_unit=1, out=self.h2)
            feat = self.feat0 if k == 0 else self.feat
            if self.merge_cls_bb0:
                st["cls_bb0"](self.h2, R, 1, 1, relu=True, m_count=count, m_unit=1, out=feat, split=(512, self.hb))
            else:
                st["cls"](self.h2, R, 1, 1, m_count=count, m_unit=1, out=feat)
            last = k == self.num_stages - 1
            if self.fuse_stage_tail and self.merge_cls_bb0 and not self.fold_deltas:
                # classifier tail + bbox_pred.2 + apply_deltas as ONE launch (two launch boundaries less per stage on the cascade's
                # chain); the last stage's launch also fuses the cascade's scores (detic_roi_heads.py:164-173)
                ops.cascade_stage_tail(feat, st["zs"], self.prob, k > 0, self.featn0 if k == 0 else None, count, R, self.C1, self.norm_temp,
                                       self.hb, st["bb2"], boxes, self.boxes[k + 1], self.cascade_weights[k], not last, float(W), float(H),
                                       zs_mem=mem_rescore[0] if (k == 0 and mem_rescore is not None) else None,
                                       prop_scores=prop_scores if (last or (k == 0 and mem_rescore is not None)) else None,
                                       mem_scores_out=mem_rescore[1] if (k == 0 and mem_rescore is not None) else None,
                                       final_inv_stages=1.0 / self.num_stages if last else 0.0, deltas_out=self.deltas)
                if k == 0 and stage0_event is not None:
                    stage0_event.record(torch.cuda.current_stream(self.device))
                boxes = self.boxes[k + 1]
                continue
            # the last stage's launch also fuses the cascade's scores: sqrt(mean_k(prob) * proposal score) (detic_roi_heads.py:164-173)
            ops.zs_classify(feat, st["zs"], self.prob, k > 0, self.featn0 if k == 0 else None, count, R, self.C1, self.norm_temp,
                            zs_mem=mem_rescore[0] if (k == 0 and mem_rescore is not None) else None,
                            prop_scores=prop_scores if (last or (k == 0 and mem_rescore is not None)) else None,
                            mem_scores_out=mem_rescore[1] if (k == 0 and mem_rescore is not None) else None,
                            final_inv_stages=1.0 / self.num_stages if last else 0.0)
            if k == 0 and stage0_event is not None:
                stage0_event.record(torch.cuda.current_stream(self.device))
            if not self.merge_cls_bb0:
                st["bb0"](self.h2, R, 1, 1, relu=True, m_count=count, m_unit=1, out=self.hb)
            st["bb2"](self.hb, R, 1, 1, m_count=count, m_unit=1, out=self.deltas)
            # next-stage proposals are clipped to the image (detic_roi_heads.py:314); the final boxes are clipped by
            # fast_rcnn_inference itself
            if self.fold_deltas and not last:
                pending = self.cascade_weights[k]
            else:
                ops.apply_deltas(self.deltas, 4, boxes, self.boxes[k + 1], count, R, self.cascade_weights[k], not last, float(W), float(H))
                boxes = self.boxes[k + 1]
        return boxes

    # ---- mask head ----------------------------------------------------------------------------------
    def forward_mask(self, views, shapes, boxes: torch.Tensor, count: torch.Tensor, cap: int, out: torch.Tensor,
                     rows: torch.Tensor = None, bufs=None, lds_reserve: int = 0, tag=None):
        """Mask head on `count` ROIs.  With `rows` (a compact ascending list of box indices) ROI k pools `boxes[rows[k]]` and its
        28x28 mask is written to `out[rows[k]]`: only the listed boxes are computed, the output layout stays per-box."""
        h3, w3 = shapes[0]
        mpool, mbuf, mup = bufs if bufs is not None else (self.mpool, self.mbuf, self.mup)
        ops.roi_align(views[0], views[1], views[2], h3, w3, 256, boxes, count, cap, 14, out=mpool, box_rows=rows)
        src, dst = mpool, mbuf
        self.deconv.lds_reserve = lds_reserve
        for conv in self.mask_convs:
            conv.lds_reserve = lds_reserve          # occupancy cap of the dense launches (EodConvDesc.lds_reserve)
            conv.event_tag = tag
            conv(src, cap, 14, 14, relu=True, m_count=count, m_unit=196, out=dst)
            src, dst = dst, src
        if self.fuse_mask_tail:
            self.deconv(src, cap, 14, 14, relu=True, m_count=count, m_unit=196, out=out, fuse=(self.pred_w, self.pred_b, rows))
        else:
            self.deconv(src, cap, 14, 14, relu=True, m_count=count, m_unit=196, out=mup)
            ops.mask_predictor_sigmoid(mup, self.pred_w, self.pred_b, cap * 784, 256, count, 784, out=out, out_units=rows)
        return out

    def forward(self, views, shapes, prop_boxes, prop_scores, prop_count, image_hw):
        """-> detections (boxes, scores, classes, rows, count) + det masks; proposals get feat/featn and masks."""
        det = self.forward_box(views, shapes, prop_boxes, prop_scores, prop_count, image_hw)
        det_boxes, det_scores, det_classes, det_rows, det_count = det
        self.forward_mask(views, shapes, det_boxes, det_count, self.topk, self.det_masks)      # forward_with_given_boxes
        return det

    def proposal_pass_buffers(self):
        """Own activation buffers for the proposal mask pass, so that it can run concurrently with the box cascade."""
        if self._prop_bufs is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            R = self.R
            self._prop_bufs = (torch.empty((R, 14, 14, 256), **f32), torch.empty((R, 14, 14, 256), **f32),
                               torch.empty((R, 28, 28, 256), **f32))
        return self._prop_bufs

    def forward_mask_memory(self, views, shapes, prop_boxes, prop_count, rows: torch.Tensor = None, rows_count: torch.Tensor = None,
                            bufs=None, tag=None):
        """`forward_mask_memory` + `mask_rcnn_inference` on ALL proposals (custom_rcnn.py:573-574).

        `rows` / `rows_count`: lazy variant -- only the proposals the memory update will read (custom_rcnn.py:875-880) get a
        mask; every other proposal's mask is dead in the reference (never read after `inference_with_proposals`)."""
        if rows is not None:
            return self.forward_mask(views, shapes, prop_boxes, rows_count, min(self.R, 128), self.prop_masks, rows=rows, bufs=bufs,
                                     lds_reserve=self.prop_pass_lds_reserve, tag=tag)
        return self.forward_mask(views, shapes, prop_boxes, prop_count, self.R, self.prop_masks, bufs=bufs,
                                 lds_reserve=self.prop_pass_lds_reserve, tag=tag)
