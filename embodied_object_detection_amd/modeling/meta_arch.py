"""`CustomRCNNRecurrent` (eval branch) on the HIP kernels: the recurrent state machine, the memory read-prep and
the memory write.

Mirrors `Detic/detic/modeling/meta_arch/custom_rcnn.py`: `forward` eval branch (435-546), `inference` (548-582),
`create_implicit_memory` (762-774), `preprocess_spatial_memory` (1019-1042), `update_implicit_memory` (681-760),
`inference_with_proposals` (825-882), `box_to_image_features` (884-901), `project_image_features` (903-936) and
`CustomRCNN._postprocess` -> detectron2 `detector_postprocess` (579-580).

Call convention: `model(batched_inputs: List[List[dict]]) -> List[dict]`; the module is stateful
between calls.  A frame is enqueued with no host synchronisation on the current HIP stream plus two scheduling streams
(box cascade + memory write; next frame's memory-independent trunk) that fork after the proposals and join before the
frame ends; the only sync per frame is the read-back of the final detection count when the result `Instances` are
materialised.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import time as _time

import numpy as np
import torch

from .. import _lib, ops
from ..checkpoint import fill_missing, load_checkpoint, load_zs_weight, reset_cls_test, synthetic_state_dict
from ..registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY
from ..structures import Boxes, Instances


# The scheduling streams are created ONCE per device and shared by every model of the process: HIP multiplexes streams onto a
# few hardware queues, and two of a model's streams landing on the same queue would silently serialise them (measured: 127 vs
# 104 frames/s for otherwise identical runs when each model created its own).  Models of one process run from one host thread,
# so sharing the streams only adds the ordering that already exists.
_SCHED_STREAMS: Dict[int, Tuple[torch.cuda.Stream, ...]] = {}
# Result sets (post-processed detections + pasted masks, detection lists): `forward` hands frame t's Instances out after frame
# t+2 has been enqueued, so that the host never waits for a detection pass that still trails on the GPU (the host needs ~3.7 ms
# to enqueue a frame, about as long as the GPU needs to run one).
RESULT_SETS = 3
PYRAMID_SETS = 6        # current frame, the frame before (its detection pass may trail), and up to four frames computed ahead


def _sched_streams(device: torch.device) -> Tuple[torch.cuda.Stream, ...]:
    """(side, look-ahead, main chain, memory selection + write): four high-priority streams.  The device offers two priority levels (0 and -1); the
    detection pass that trails under the next frame runs at 0, everything latency-bound at -1."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _SCHED_STREAMS:
        _SCHED_STREAMS[idx] = tuple(torch.cuda.Stream(device=device, priority=-1) for _ in range(4))
    return _SCHED_STREAMS[idx]


_DET_STREAMS: Dict[Tuple[int, int], torch.cuda.Stream] = {}


def _det_stream(device: torch.device, priority: int) -> torch.cuda.Stream:
    """The detection-pass stream, also ONE per device and process.  A stream per model instance made the frame rate of otherwise
    identical models bimodal (270 or 200-220 frames/s at 640x640, tools/knob_ab.py): the runtime hands hardware queues out round
    robin (GPU_MAX_HW_QUEUES), and a later model's stream could land on the queue of one of the chain streams, which serialises
    the detection pass with that chain."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, int(priority))
    if key not in _DET_STREAMS:
        _DET_STREAMS[key] = torch.cuda.Stream(device=device, priority=0 if int(priority) >= 0 else -1)
    return _DET_STREAMS[key]


@META_ARCH_REGISTRY.register()
class CustomRCNNRecurrent:
    def __init__(self, cfg, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        dev = str(cfg.MODEL.DEVICE)
        if not dev.startswith("cuda"):
            raise _lib.EodError(
                f"MODEL.DEVICE={dev!r}: the product path is HIP-only (no CPU fallback). The CPU restatement lives in "
                "oracle/ and is test infrastructure.")
        if not torch.cuda.is_available():
            raise _lib.EodError("no GPU visible: the HIP product path cannot run")
        _lib.load()
        self.cfg = cfg
        self.device = torch.device(dev if ":" in dev else "cuda:0")
        torch.cuda.set_device(self.device)
        self.map_conditioned = cfg.MODEL.TIMM.BASE_NAME == "resnet50_in21k_map"
        self.memory_type = cfg.MODEL.MEMORY_TYPE
        self.save_semmap = bool(cfg.MODEL.TEST_SAVE_SEMMAP)
        self.output_dir = cfg.OUTPUT_DIR
        self.cls_score_thresh = float(cfg.MODEL.MEMORY_CLS_SCORE_THRESH)
        self.obs_score_thresh = float(cfg.MODEL.MEMORY_OBS_SCORE_THRESH)
        self.test_type = cfg.MODEL.TEST_TYPE
        if self.test_type not in ("default", "episodic", "longterm"):
            raise ValueError(f"MODEL.TEST_TYPE={self.test_type!r}")
        self.pixel_mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self.pixel_std = [float(v) for v in cfg.MODEL.PIXEL_STD]
        self.mask_threshold = 0.5
        self.training = False
        # the memory write-back and the 256-proposal mask pass run for every MEMORY_TYPE in the reference
        # (custom_rcnn.py:515,573); keep that for like-for-like timing
        self.always_update_memory = True
        # Default: compute the proposal masks only for the proposals the memory update reads (<= 100 unique rows kept by
        # `inference_with_proposals`, custom_rcnn.py:875-880).  The reference runs the mask head on all 256 proposals
        # (custom_rcnn.py:573) and never reads the others; they are not observable through the boundary and every output is
        # bitwise identical (tests/test_model_gpu.py::test_lazy_proposal_masks_give_identical_results).  `False` restores the
        # reference-faithful 256-proposal pass (bench.py reports it beside the headline).
        self.lazy_proposal_masks = True
        # Detections that come from one proposal carry the SAME class-agnostic box (fast_rcnn_inference keeps up to 20 classes of a
        # proposal as separate detections, detic_roi_heads.py:214-221) and the mask head is class agnostic (CLS_AGNOSTIC_MASK): their
        # masks are identical.  Default: the mask head runs once per distinct box (~100 of 300 detections on the benchmark scene)
        # and every detection of the group reads that mask; outputs are bitwise those of the 300-ROI pass
        # (tests/test_model_gpu.py::test_detection_mask_groups_give_identical_results).  `False` restores the reference-faithful
        # one-ROI-per-detection pass (bench.py reports it beside the headline).
        self.dedup_detection_masks = True
        # The proposal mask pass (custom_rcnn.py:573) needs only the proposals and the FPN features, not the box cascade: the
        # cascade's small latency-bound launches (15 FC GEMMs, 3 ROIAligns, the selection sorts) are enqueued on a second,
        # high-priority HIP stream and run beside the proposal pass's large GEMMs.  The detection mask pass follows on the main
        # stream after both (the two large passes never share the chip).  Same kernels, same inputs, same results;
        # `overlap_branches = False` restores one stream.
        self.overlap_branches = True
        self._side_stream = None
        self._ev_props = self._ev_pm = self._ev_box = self._ev_mem = self._ev_sel = self._ev_s0 = None
        # Look-ahead: the ResNet trunk does not read the memory, so the NEXT frame's bottom-up pass and FPN top-down convs (known from the
        # inner frame list of `forward`, or passed as `next_frame`) are enqueued on a third stream while this frame's mask passes
        # run; they write the other of two pyramid buffer sets.
        self.prefetch_trunk = True
        self.lookahead_at_start = True   # start it at the beginning of the frame (True) or once the proposals exist (False)
        self._ev_start = None
        self._trunk_stream = None
        self._ev_trunk = None
        self._prefetched = None      # (image object of the frame, padded H, W)
        self._pyramid = 0            # which of the two FPN buffer sets the current frame uses
        self.overlap_memory_write = True    # also the memory selection + write-back, beside the detection mask pass
        # Cross-frame pipelining inside `forward([episode])`: nothing of frame t+1 depends on frame t's DETECTION masks (the
        # memory write needs the proposal masks only), so the detection mask pass + post-processing + paste of frame t run on
        # their own low-priority stream underneath frame t+1's memory read, tower, proposal decoding and box cascade (short
        # latency-bound chains that leave most of the chip idle).  Three pyramid sets and two detection-list sets make the
        # overlap hazard free; results are bitwise those of the in-order schedule.
        self.pipeline_detection_pass = True
        # may the detection pass of frame t still run when frame t+1 starts?  (False: the frame's chain joins it at the end of the frame)
        self.trail_detection_pass = True
        self.det_stream_priority = 0         # 0 = normal, -1 = high (like the chains)
        self.trunk_stream_priority = -1      # -1 = high (default), 0 = a normal-priority stream of its own
        # Memory selection right after cascade stage 0 on its own stream (it needs only the stage-0 features) instead of after the
        # cascade.  Measured (tools/frame_schedule.py, same box): True lets the proposal-mask pass and the memory write finish
        # early, the next frame then starts while the detection pass still runs and its latency-bound chain is slowed 3x by the
        # resident GEMM workgroups (5.66 ms/frame); False keeps the detection pass inside its frame (5.52 ms/frame).  A CU mask
        # on the detection stream (hipExtStreamCreateWithCUMask) would be the remedy; this runtime accepts the call and ignores
        # the mask (an fp32 matmul on a half-masked stream takes the same time).
        self.early_memory_selection = False
        self.memory_selection_first = True     # on the side stream: cascade -> memory selection -> detection selection
        # how many coming frames of an episode the look-ahead computes at once (their images are all there when `forward` is
        # called): 2 = the memory-independent trunk + FPN top-down of frames t+1 and t+2 as ONE N = 2 pass every second frame
        # (planned like one image: bitwise the N = 1 results).  Measured at 640x640: the pass costs 1.23 ms per image instead of 1.61
        # (tools/trunk_batch_time.py).  With a window of the same size (round 3) the frames WITH the double trunk lost what the
        # others gained; inside a deeper window (`lookahead_depth`) the N = 2 pass has two frame periods to finish: default 2.
        self.lookahead_frames = 2
        # how many coming frames may be AHEAD at any time (>= lookahead_frames; PYRAMID_SETS - 2 at most).  With depth == batch size
        # a pass starts when the last frame computed ahead has been consumed and must be over one frame later -- the trunk's ~75
        # launches then span the whole frame (profiles/r03_frame_schedule.txt: 0.24 -> 3.60 ms of a 3.5 ms frame) and the next frame
        # waits for them: the look-ahead is a second critical chain.  A deeper window decouples it: a pass is started while frames
        # are still ahead and has several frame periods to finish (only its throughput matters, and N = 2 passes cost 24 % less per
        # image).  None = lookahead_frames (the round-3 schedule).  Measured at 640x640 (tools/knob_ab.py, one call): batch 1 /
        # window 1 287 frames/s, 1 / 2 287, 2 / 3 301, 2 / 4 300, 3 / 5 297, 4 / 6 295 (the last two with eight pyramid sets).
        self.lookahead_depth: Optional[int] = 3
        self._ahead: List[dict] = []        # frames computed (or being computed) ahead, in order: image, Hp, Wp, event
        self.front_event: Optional[torch.cuda.Event] = None      # one-stream schedule only: recorded after the box cascade
        # where the (deferred, low-priority) detection mask pass may start: "cascade" (as soon as the detections exist),
        # "proposal_masks" / "memory_write" (behind this frame's critical chain: it then runs beside the NEXT frame's
        # latency-bound front instead of beside this frame's proposal masks and memory write)
        self.detection_pass_after = "cascade"
        self._det_stream = None
        self._ev_call = None
        self._ev_det = [None] * RESULT_SETS   # per result set: detection pass + paste finished
        self._pyr_reader = {}                # pyramid set -> event of the last detection pass that read it
        self._frame_no = 0

        num_classes = int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        if state_dict is None:
            if cfg.MODEL.WEIGHTS and str(cfg.MODEL.WEIGHTS).endswith((".pth", ".pkl")) and __import__("os").path.exists(cfg.MODEL.WEIGHTS):
                sd, _ = load_checkpoint(cfg.MODEL.WEIGHTS, num_classes)
                state_dict = fill_missing(sd, 0, num_classes)
            else:
                state_dict = synthetic_state_dict(0, num_classes, cfg.MODEL.ROI_BOX_HEAD.ZEROSHOT_WEIGHT_PATH)
        if cfg.MODEL.RESET_CLS_TESTS and cfg.MODEL.TEST_CLASSIFIERS:
            import os
            p = cfg.MODEL.TEST_CLASSIFIERS[0]
            if not os.path.exists(p):
                p = cfg.MODEL.ROI_BOX_HEAD.ZEROSHOT_WEIGHT_PATH
            state_dict = dict(state_dict)
            reset_cls_test(state_dict, p, int(cfg.MODEL.TEST_NUM_CLASSES[0]))          # utils.py:32-50
        self.state_dict_ref = state_dict
        # zs_weight of the meta-arch itself (custom_rcnn.py:375-382)
        self.zs_weight = load_zs_weight(cfg.MODEL.ROI_BOX_HEAD.ZEROSHOT_WEIGHT_PATH).contiguous().to(self.device)
        self.C1 = self.zs_weight.shape[1]

        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, state_dict, self.device)
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg, state_dict, self.device)
        self.roi_heads = ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, state_dict, self.device, self.proposal_generator.cap)
        R = self.proposal_generator.cap
        self.mem_scores = torch.zeros((R, self.C1), dtype=torch.float32, device=self.device)
        # inference_with_proposals' selection (custom_rcnn.py:862-875) in one launch: threshold, per-class NMS, top 100, unique rows
        self.mem_selector = ops.DetectionSelector(R, self.C1, 100, self.device, unique=True)
        self._uniq_rows, self._uniq_count = self.mem_selector.uniq_rows, self.mem_selector.uniq_count
        self._mem_scores_frame = -1      # frame whose CLIP re-score `mem_scores` holds (written by stage 0 of the cascade)
        # recurrent state
        self.implicit_memory: Optional[torch.Tensor] = None   # [N,512] f32  (== semmap_features)
        self.observations: Optional[torch.Tensor] = None      # [N] f32      (== observation_count)
        self.semmap = None
        self._mem_f16: Optional[torch.Tensor] = None
        self._dirty: Optional[torch.Tensor] = None          # int32 [N]: rows of the fp16 snapshot that are out of date
        self._f16_valid = False
        self._dirty_pending = False          # some write marked rows in `_dirty` that the snapshot does not have yet
        # TEST_TYPE default / episodic read the memory as it stands at every frame (loader.py:289-293): the write refreshes the
        # snapshot rows it invalidates itself (ops.MemoryWriter `snapshot`), no normalise launch at the start of the next frame.
        # `longterm` freezes the snapshot after the first frame: there the write only marks the rows.
        self.snapshot_follows_write: Optional[bool] = None      # None: decided by TEST_TYPE at every write
        self._err = torch.zeros((1,), dtype=torch.int32, device=self.device)   # device error word (EOD_FLAG_*), read with the count
        self._writer = None
        self._writer_key = None
        self._post = None
        self._posts = None
        self._post_slot = 0
        self.last_stats: Dict[str, torch.Tensor] = {}
        self.stats_log = None
        self.trace = None
        self.host_profile = {"frames": 0, "enqueue_s": 0.0, "materialize_s": 0.0, "wait_s": 0.0}   # host seconds spent in forward()

    # detectron2 nn.Module surface used by the drivers
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        """Training mode: `forward` runs `forward_model` on every frame through the attached `modeling.training.Trainer`."""
        self.training = bool(mode)
        return self

    def to(self, *_a, **_k):
        return self

    def __call__(self, batched_inputs):
        return self.forward(batched_inputs)

    # ---- state ----------------------------------------------------------------------------------------
    def reset_memory(self, n_cells: int):
        """`frame['memory_reset']` branch (custom_rcnn.py:470-479)."""
        if self.implicit_memory is None or self.implicit_memory.shape[0] != n_cells:
            self.implicit_memory = torch.empty((n_cells, 512), dtype=torch.float32, device=self.device)
            self.observations = torch.empty((n_cells,), dtype=torch.float32, device=self.device)
            self._mem_f16 = torch.empty((n_cells, 512), dtype=torch.float16, device=self.device)
            self._dirty = torch.empty((n_cells,), dtype=torch.int32, device=self.device)
        lib = _lib.load()
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.eod_fill_f32(self.implicit_memory.data_ptr(), 0.0, self.implicit_memory.numel(), s), "fill")
        _lib.check(lib.eod_fill_f32(self.observations.data_ptr(), 0.0, self.observations.numel(), s), "fill")
        # the resident fp16 snapshot of an all-zero memory is all zero; from here on it is kept incrementally (dirty rows only)
        _lib.check(lib.eod_fill_i32(self._mem_f16.data_ptr(), 0, self._mem_f16.numel() // 2, s), "fill")
        _lib.check(lib.eod_fill_i32(self._dirty.data_ptr(), 0, n_cells, s), "fill")
        self._f16_valid = True
        self._dirty_pending = False
        self.semmap = None

    def invalidate_memory_snapshot(self):
        """Call after writing `implicit_memory` / `observations` from outside (e.g. a loaded snapshot): the next frame
        re-normalises the whole table instead of the dirty rows."""
        self._f16_valid = False

    def _refresh_memory_snapshot(self):
        """a4 + fp16 cast (create_implicit_memory + preprocess_spatial_memory): bring the resident fp16 table up to the state."""
        if self._f16_valid:
            if self._dirty_pending:
                ops.memory_normalize_dirty_f16(self.implicit_memory, self.observations, self._dirty, self._mem_f16)
        else:
            ops.memory_normalize_f16(self.implicit_memory, self.observations, out=self._mem_f16)
            _lib.check(_lib.load().eod_fill_i32(self._dirty.data_ptr(), 0, self._dirty.numel(), torch.cuda.current_stream().cuda_stream), "fill")
            self._f16_valid = True
        self._dirty_pending = False

    def _ensure_frame_buffers(self, H: int, W: int, n_cells: int):
        key = (H, W, n_cells)
        if self._writer_key != key:
            self._writer = ops.MemoryWriter(H, W, n_cells, 100, self.proposal_generator.cap, self.device, mask_thresh=0.5)
            self._writer_key = key
        if self._posts is None or self._posts[0]["hw"] != (H, W):
            # RESULT_SETS result sets: frame t's `Instances` are sliced out of set t % 3 only after frame t+2 has been enqueued
            # (`forward`), so the host's read-back of the detection count never leaves the GPU idle
            D = self.roi_heads.topk
            dev = self.device
            # boxes / scores / classes / masks of a set are allocated anew for every frame (`_postprocess_and_paste`) and handed to the
            # caller as they are: no copy of the 0/1 byte masks ([300,H,W]: 123 MB at 640x640) when the Instances are built
            self._posts = [dict(
                hw=(H, W), boxes=None, scores=None, classes=None, masks=None,
                src=torch.zeros((D,), dtype=torch.int32, device=dev), count=torch.zeros((1,), dtype=torch.int32, device=dev),
                count_host=torch.zeros((1,), dtype=torch.int32).pin_memory(), err_host=torch.zeros((1,), dtype=torch.int32).pin_memory(),
                ready=torch.cuda.Event(), err_ready=torch.cuda.Event()) for _ in range(RESULT_SETS)]
            self._post_slot = 0
        self._post = self._posts[self._post_slot]

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, batched_inputs: List[List[dict]]):
        """Sequential pass over sequences and frames; the memory persists across calls (custom_rcnn.py:435-546)."""
        if self.training:
            # custom_rcnn.py:444-461: forward_model per frame, the loss terms summed.  There is no autograd graph on this path: the
            # attached `modeling.training.Trainer` runs forward AND backward and keeps the gradients for its `optimizer_step()`.
            if getattr(self, "trainer", None) is None:
                raise RuntimeError("training mode needs a trainer: `modeling.training.Trainer(model, state_dict)` attaches itself to the "
                                   "model; then `model.train(); losses = model(data); trainer.optimizer_step()`")
            return self.trainer.forward_backward_frames(batched_inputs)
        if self.overlap_branches and self.pipeline_detection_pass:
            # The frame's own chain (memory read -> tower -> proposals -> proposal masks -> memory write) moves to a HIGH priority
            # stream for the duration of the call: the previous frame's detection pass trails at normal priority, and at equal
            # priority the short chain would queue behind the GEMMs' thousands of workgroups.  The caller's stream is joined on
            # both sides, so the call keeps its in-order meaning.
            caller = torch.cuda.current_stream(self.device)
            ms = _sched_streams(self.device)[2]
            if self._ev_call is None:
                self._ev_call = (torch.cuda.Event(), torch.cuda.Event())
            self._ev_call[0].record(caller)
            ms.wait_event(self._ev_call[0])
            with torch.cuda.stream(ms):
                out = self._forward_frames(batched_inputs)
                self._ev_call[1].record(ms)
            caller.wait_event(self._ev_call[1])
            return out
        return self._forward_frames(batched_inputs)

    def _forward_frames(self, batched_inputs: List[List[dict]]):
        batch_output = []
        pending = []              # tickets of the last frames: Instances are built RESULT_SETS - 1 frames behind the enqueue
        for input_seq in batched_inputs:
            for i, frame in enumerate(input_seq):
                n_cells = int(input_seq[0]["memory"].shape[0])
                if frame["memory_reset"]:
                    self.reset_memory(n_cells)
                if self.implicit_memory is None:
                    raise RuntimeError("first frame of a scene must carry memory_reset=True (custom_rcnn.py:485 reads unset state)")
                refresh = self.test_type in ("default", "episodic") or (self.test_type == "longterm" and i == 0)
                nxt = input_seq[i + 1:i + 1 + PYRAMID_SETS] or None                            # the frames that follow, in order
                last = nxt is None and input_seq is batched_inputs[-1]
                t0 = _time.perf_counter()
                self.inference_frame(frame, refresh_memory_snapshot=refresh, materialize=False, next_frame=nxt,
                                     trailing_detection_pass=(not last) and self.trail_detection_pass)
                pending.append(self._post_ticket())
                t1 = _time.perf_counter()
                if len(pending) == RESULT_SETS:
                    batch_output.append({"instances": self._materialize(pending.pop(0))})
                hp = self.host_profile
                hp["frames"] += 1
                hp["enqueue_s"] += t1 - t0
                hp["materialize_s"] += _time.perf_counter() - t1
                if self.save_semmap and i == 0:
                    self.save_memory_snapshot(frame["sequence_name"])          # custom_rcnn.py:518-530
        for ticket in pending:
            batch_output.append({"instances": self._materialize(ticket)})
        return batch_output

    def _post_ticket(self):
        """Async read-back of the frame's detection count into pinned host memory + an event; flips the result set."""
        P = self._post
        cur = torch.cuda.current_stream(self.device)
        P["err_host"].copy_(self._err, non_blocking=True)
        k = self._post_slot
        if self.overlap_branches and self.pipeline_detection_pass and self._det_stream is not None and self._ev_det[k] is not None:
            # the detection pass of this frame may still be running on its own stream: the count is copied there, and the ticket's
            # event covers both streams
            ds = self._det_stream
            ds.wait_event(self._ev_det[k])
            P["err_ready"].record(cur)
            ds.wait_event(P["err_ready"])
            with torch.cuda.stream(ds):
                P["count_host"].copy_(P["count"], non_blocking=True)
                P["ready"].record(ds)
        else:
            P["count_host"].copy_(P["count"], non_blocking=True)
            P["ready"].record(cur)
        self._post_slot = (self._post_slot + 1) % RESULT_SETS
        return P

    def _device_image(self, frame) -> torch.Tensor:
        img = frame["image"]
        if not torch.is_tensor(img):
            img = torch.as_tensor(np.asarray(img))
        if img.dtype != torch.uint8:
            img = img.to(torch.uint8)
        return img.to(self.device, non_blocking=True).contiguous()

    def _device_proj(self, frame) -> torch.Tensor:
        p = frame["proj_indices"]
        if not torch.is_tensor(p):
            p = torch.from_numpy(np.ascontiguousarray(p))
        if p.dim() == 3:
            p = p.squeeze(2)
        if p.dtype != torch.int32:
            p = p.to(torch.int32)
        return p.to(self.device, non_blocking=True).contiguous()

    def _enqueue_trunk(self, frames: List[dict], after: torch.cuda.Event, first: int = 0):
        """Bottom-up pass + FPN top-down convs of the coming frame(s) on the look-ahead stream, ordered after `after` (an event of
        the main stream recorded once the previous frame has fully finished).  Frame b of the list is written into pyramid set
        (current + 1 + first + b) -- `first` frames are ahead already; more than one frame = one batched pass.  The frames join
        `self._ahead` with the pass's own event."""
        if self._trunk_stream is None:
            # high priority like the side stream: its ~75 launches are small and must not queue behind the mask GEMMs' thousands
            # of workgroups
            self._trunk_stream = _sched_streams(self.device)[1] if self.trunk_stream_priority < 0 else _det_stream(self.device, 1000)
            self._ev_trunk = torch.cuda.Event()
        ts = self._trunk_stream
        ts.wait_event(after)
        done = torch.cuda.Event()
        sets = [(self._pyramid + 1 + first + b) % PYRAMID_SETS for b in range(len(frames))]
        for st in sets:
            if st in self._pyr_reader:
                ts.wait_event(self._pyr_reader[st])      # a trailing detection pass may still read that set
        with torch.cuda.stream(ts):
            self._mark("trunk_lookahead_begin", ts)
            xs = []
            for f in frames:
                x4, Hp, Wp = ops.preprocess_image(self._device_image(f), self.pixel_mean, self.pixel_std)
                xs.append(x4)
            if len(frames) == 1:
                self.backbone.top_down(self.backbone.bottom_up.forward(xs[0], Hp, Wp), Hp, Wp, sets[0])
            else:
                n = len(frames)
                c = self.backbone.bottom_up.forward(torch.cat(xs, dim=0), Hp, Wp, N=n)
                p345 = self.backbone.top_down_batched(c, Hp, Wp, n)
                for b, st in enumerate(sets):
                    views = self.backbone._plan(Hp, Wp, st)[3]
                    for l in range(3):
                        views[l].copy_(p345[l][b:b + 1])
            done.record(ts)
            self._mark("trunk_lookahead", ts)
        self._ahead = self._ahead + [dict(image=f["image"], Hp=Hp, Wp=Wp, event=done) for f in frames]

    def inference_frame(self, frame: dict, refresh_memory_snapshot: bool = True, materialize: bool = True,
                        next_frame=None, trailing_detection_pass: bool = False):
        """One frame: `inference` (custom_rcnn.py:548-582) + `update_implicit_memory` (681-760).  `next_frame` (optional) is the
        frame the caller will pass next -- or the list of the frames it will pass next, in order: their memory-independent
        bottom-up passes are started early (`lookahead_frames` of them at a time as one batched pass).  `trailing_detection_pass`
        (set by `forward` for every frame but the last of a call): do not join the detection-pass stream at the end of the frame;
        the caller reads the results through `_post_ticket` / `_materialize` only."""
        H, W = int(frame["image"].shape[-2]), int(frame["image"].shape[-1])
        if H % 32 or W % 32:
            raise ValueError("H and W must be multiples of 32 (proj_indices is not padded: SURVEY §8 notation)")
        proj = self._device_proj(frame)
        if tuple(proj.shape) != (H, W):
            raise ValueError(f"proj_indices shape {tuple(proj.shape)} != image {(H, W)}")
        n_cells = self.implicit_memory.shape[0]
        self._ensure_frame_buffers(H, W, n_cells)
        self._frame_no += 1
        self._mark("start")

        # a4 + fp16 cast (create_implicit_memory + preprocess_spatial_memory)
        mem_f16 = None
        if self.memory_type == "implicit_memory":
            if refresh_memory_snapshot:
                self._refresh_memory_snapshot()
            mem_f16 = self._mem_f16

        cur = torch.cuda.current_stream(self.device)
        ext, self._prefetched = self._prefetched, None
        if isinstance(ext, tuple):                 # (image, Hp, Wp) computed into the next set by BatchedSequences, its event in _ev_trunk
            self._ahead = [dict(image=ext[0], Hp=ext[1], Wp=ext[2], event=self._ev_trunk)]
        ahead = self._ahead
        hit = bool(ahead) and ahead[0]["image"] is frame["image"]
        if ahead and not hit:
            for e in ahead:                        # the hint was wrong: what runs ahead must be over before any set or the trunk's
                cur.wait_event(e["event"])         # own buffers are touched again
            ahead = []
        pre = None
        if hit:
            pre = ahead.pop(0)
            cur.wait_event(pre["event"])           # this frame's own pass only: later passes may still be running
        # every frame moves to the next of PYRAMID_SETS pyramid sets (the look-ahead wrote P3..P5 of a hit into exactly that one);
        # a trailing detection pass of the frame that last used the set must be over before a miss recomputes into it
        self._pyramid = (self._pyramid + 1) % PYRAMID_SETS
        if not hit and self._pyramid in self._pyr_reader:
            cur.wait_event(self._pyr_reader[self._pyramid])
        coming = [] if next_frame is None else (list(next_frame) if isinstance(next_frame, (list, tuple)) else [next_frame])
        # frames further ahead stay valid only if the caller really passes them next (checked frame by frame)
        keep = 0
        while keep < len(ahead) and keep < len(coming) and ahead[keep]["image"] is coming[keep]["image"]:
            keep += 1
        for e in ahead[keep:]:
            cur.wait_event(e["event"])             # dropped: over before their sets are written again
        ahead = ahead[:keep]
        self._ahead = ahead
        depth = max(1, int(self.lookahead_frames)) if self.lookahead_depth is None else max(1, int(self.lookahead_depth))
        batch = []
        if self.prefetch_trunk and self.overlap_branches and len(ahead) < min(depth, PYRAMID_SETS - 2):
            room = PYRAMID_SETS - 2 - len(ahead)
            batch = coming[len(ahead):len(ahead) + min(max(1, int(self.lookahead_frames)), room)]
            # one pass = one image size (the N = 2 pass stacks the images); a frame of another size starts its own pass later
            n_same = 0
            for f in batch:
                if tuple(f["image"].shape) != tuple(batch[0]["image"].shape):
                    break
                n_same += 1
            batch = batch[:n_same]
        next_frame = batch if batch else None
        look_ahead = next_frame is not None
        first_ahead = len(ahead)
        if look_ahead and self.lookahead_at_start:
            # it may start NOW, beside this frame's memory fusion, tower and proposal decoding (a short latency-bound chain
            # that leaves most of the chip idle); the host enqueues that chain first so that the main stream never starves
            if self._ev_start is None:
                self._ev_start = torch.cuda.Event()
            self._ev_start.record(cur)
        if hit:
            feats, views, shapes, off = self.backbone.fuse_memory_and_top(pre["Hp"], pre["Wp"], mem_f16, proj, self._pyramid, self._err)
        else:
            x4, Hp, Wp = ops.preprocess_image(self._device_image(frame), self.pixel_mean, self.pixel_std)
            feats, views, shapes, off = self.backbone.forward(x4, Hp, Wp, mem_f16, proj, self._pyramid, self._err)
        prop_boxes, prop_scores, prop_count = self.proposal_generator.forward(feats, shapes, off)
        if look_ahead and self.lookahead_at_start:
            self._enqueue_trunk(next_frame, self._ev_start, first_ahead)
        update_mem = self.memory_type == "implicit_memory" or self.always_update_memory
        mem_sel = None
        mem_done = False
        if self.overlap_branches:
            main = torch.cuda.current_stream(self.device)
            if self._side_stream is None:
                self._side_stream = _sched_streams(self.device)[0]     # high priority: the small launches go first
                self._ev_props, self._ev_pm, self._ev_box, self._ev_mem, self._ev_sel, self._ev_s0 = (torch.cuda.Event() for _ in range(6))
            self._ev_props.record(main)
            self._mark("proposals", main)
            if look_ahead and not self.lookahead_at_start:
                self._enqueue_trunk(next_frame, self._ev_props, first_ahead)
            lazy = self.lazy_proposal_masks and update_mem
            # Host enqueue order matters (the GPU runs behind the host here): first the large launches of the main stream, then
            # the side stream's ~45 small ones -- they all execute beside the two mask passes.
            if not lazy:
                prop_masks = self.roi_heads.forward_mask_memory(views, shapes, prop_boxes, prop_count,
                                                                bufs=self.roi_heads.proposal_pass_buffers(), tag=("prop_all", self._frame_no))
                self._ev_pm.record(main)
            self._side_stream.wait_event(self._ev_props)
            with torch.cuda.stream(self._side_stream):
                k = self._post_slot
                if self._ev_det[k] is not None:
                    self._side_stream.wait_event(self._ev_det[k])     # the detection list set k is still read by frame t-2's pass
                # The frame's critical chain (proposal masks -> memory write -> next frame's memory read) waits for the MEMORY
                # selection, which needs the cascade's stage-0 features only; the detection selection feeds the detection pass,
                # which has slack.  Same stream, memory selection first (`memory_selection_first`).
                sel_first = lazy and not self.early_memory_selection and self.memory_selection_first
                box_sel = {}

                def _select_memory():
                    box_sel["mem"] = self.select_memory_instances(prop_boxes, prop_scores, prop_count, (H, W))
                    self._ev_sel.record(self._side_stream)
                    self._mark("mem_select", self._side_stream)

                if update_mem:
                    self._mem_scores_frame = self._frame_no          # written by stage 0 of the cascade below
                det = self.roi_heads.forward_box(views, shapes, prop_boxes, prop_scores, prop_count, (H, W), sel=k,
                                                 stage0_event=self._ev_s0 if (lazy and self.early_memory_selection) else None,
                                                 mem_rescore=(self.zs_weight, self.mem_scores) if update_mem else None,
                                                 after_cascade=_select_memory if sel_first else None)
                det_boxes, det_scores, det_classes, det_rows, det_count = det
                self._ev_box.record(self._side_stream)
                self._mark("cascade+det_select", self._side_stream)
                if sel_first:
                    mem_sel = box_sel["mem"]
            mem_stream = self._side_stream
            if lazy and not self.early_memory_selection and not self.memory_selection_first:
                with torch.cuda.stream(self._side_stream):
                    mem_sel = self.select_memory_instances(prop_boxes, prop_scores, prop_count, (H, W))
                    self._ev_sel.record(self._side_stream)
                    self._mark("mem_select", self._side_stream)
            elif lazy and not self.early_memory_selection:
                pass
            elif lazy:
                # The memory selection needs only stage 0 of the cascade (its CLIP-space features, custom_rcnn.py:825-875): it
                # runs on its own stream beside stages 1-2 and the detection selection; the mask head then runs only on the
                # proposals it keeps (same results: the other proposals' masks are never read, custom_rcnn.py:875-880).
                mem_stream = _sched_streams(self.device)[3]
                mem_stream.wait_event(self._ev_s0)
                with torch.cuda.stream(mem_stream):
                    mem_sel = self.select_memory_instances(prop_boxes, prop_scores, prop_count, (H, W))
                    self._ev_sel.record(mem_stream)
                    self._mark("mem_select", mem_stream)
            pipelined = self.pipeline_detection_pass
            det_after = self.detection_pass_after if (pipelined and lazy and update_mem and self.overlap_memory_write) else "cascade"
            if pipelined and det_after == "cascade":
                self._enqueue_detection_pass(views, shapes, det, (H, W), frame)
            elif pipelined:
                pass        # enqueued below, behind the proposal masks / the memory write
            else:
                main.wait_event(self._ev_box)
                self._detection_masks(views, shapes, det_boxes, det_count)
            if lazy:
                main.wait_event(self._ev_sel)
                prop_masks = self.roi_heads.forward_mask_memory(views, shapes, prop_boxes, prop_count, rows=self._uniq_rows,
                                                                rows_count=self._uniq_count,
                                                                bufs=self.roi_heads.proposal_pass_buffers(), tag=("prop", self._frame_no))
                self._ev_pm.record(main)
                self._mark("prop_masks", main)
            if pipelined and det_after == "proposal_masks":
                self._enqueue_detection_pass(views, shapes, det, (H, W), frame, after=self._ev_pm)
            if update_mem and self.overlap_memory_write:
                # the memory write needs the proposal masks (main stream) and the selection (side stream): it runs on the side
                # stream beside the detection mask pass; the main stream joins at the end of the frame
                with torch.cuda.stream(mem_stream):
                    if not lazy:
                        mem_sel = self.select_memory_instances(prop_boxes, prop_scores, prop_count, (H, W))
                    mem_stream.wait_event(self._ev_pm)
                    self.update_implicit_memory(prop_boxes, prop_scores, prop_count, prop_masks, proj, (H, W), mem_sel)
                    self._ev_mem.record(mem_stream)
                    self._mark("mem_write", mem_stream)
                mem_done = True
                if pipelined and det_after == "memory_write":
                    self._enqueue_detection_pass(views, shapes, det, (H, W), frame, after=self._ev_mem)
        else:
            pipelined = False
            det = self.roi_heads.forward_box(views, shapes, prop_boxes, prop_scores, prop_count, (H, W),
                                             mem_rescore=(self.zs_weight, self.mem_scores) if update_mem else None)
            if update_mem:
                self._mem_scores_frame = self._frame_no
            det_boxes, det_scores, det_classes, det_rows, det_count = det
            if self.front_event is not None:
                # the latency-bound front of the frame (memory fusion, tower, proposal decoding, cascade) ends here; what follows is
                # dense (mask passes): BatchedSequences staggers its scenes on this point
                self.front_event.record(torch.cuda.current_stream(self.device))
            self._detection_masks(views, shapes, det_boxes, det_count)
            if self.lazy_proposal_masks and update_mem:
                # select the memory instances first, then run the mask head only on those proposals (same results: the other
                # proposals' masks are never read, custom_rcnn.py:875-880)
                mem_sel = self.select_memory_instances(prop_boxes, prop_scores, prop_count, (H, W))
                prop_masks = self.roi_heads.forward_mask_memory(views, shapes, prop_boxes, prop_count, rows=self._uniq_rows,
                                                                rows_count=self._uniq_count, tag=("prop", self._frame_no))
            else:
                prop_masks = self.roi_heads.forward_mask_memory(views, shapes, prop_boxes, prop_count, tag=("prop_all", self._frame_no))

        # detector_postprocess (custom_rcnn.py:579-580)
        P = self._post
        if not pipelined:
            self._postprocess_and_paste(det_boxes, det_scores, det_classes, det_count, (H, W), frame, P)

        # memory update (custom_rcnn.py:515)
        if mem_done:
            torch.cuda.current_stream(self.device).wait_event(self._ev_mem)
        elif update_mem:
            self.update_implicit_memory(prop_boxes, prop_scores, prop_count, prop_masks, proj, (H, W), mem_sel)
        self.last_stats = {"prop_count": prop_count, "det_count": P["count"], "mem_k": self._writer.k_out,
                           "det_mask_rois": self.roi_heads.last_selector.rep_count if self.dedup_detection_masks else P["count"]}
        if pipelined and not trailing_detection_pass:
            torch.cuda.current_stream(self.device).wait_event(self._ev_det[self._post_slot])     # in-order callers see a finished frame
        if self.stats_log is not None:      # bench.py: device-side copies of the frame's counters, read after the timed region
            if pipelined and trailing_detection_pass:
                with torch.cuda.stream(self._det_stream):      # the count is written by the trailing detection pass: copy it there
                    cnt = P["count"].clone()
            else:
                cnt = P["count"].clone()
            self.stats_log.append((prop_count.clone(), cnt, self._writer.k_out.clone(), self._uniq_count.clone(),
                                   self.last_stats["det_mask_rois"].clone()))
        if not materialize:
            return None
        return {"instances": self._materialize(self._post_ticket())}

    def _mark(self, name: str, stream=None):
        """Diagnostics (`model.trace = []`): a timing event on `stream` (default: current) at a named point of the frame's schedule;
        tools/frame_schedule.py turns the list into a per-frame timeline without a profiler in the way."""
        if self.trace is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream(self.device))
        self.trace.append((self._frame_no, name, ev))

    def _detection_masks(self, views, shapes, det_boxes, det_count):
        """`forward_with_given_boxes` (detic_roi_heads.py:257): the mask head on the detections -- once per distinct box when
        `dedup_detection_masks` (the groups come from the detection selection's launch)."""
        rh = self.roi_heads
        sel = rh.last_selector
        if self.dedup_detection_masks:
            return rh.forward_mask(views, shapes, det_boxes, sel.rep_count, rh.topk, rh.det_masks, rows=sel.rep_list,
                                   lds_reserve=rh.det_pass_lds_reserve, tag=("det", self._frame_no))
        return rh.forward_mask(views, shapes, det_boxes, det_count, rh.topk, rh.det_masks, lds_reserve=rh.det_pass_lds_reserve,
                               tag=("det", self._frame_no))

    def _postprocess_and_paste(self, det_boxes, det_scores, det_classes, det_count, image_hw, frame, P):
        H, W = image_hw
        out_h, out_w = int(frame.get("height", H)), int(frame.get("width", W))
        if (out_h, out_w) != (H, W):
            raise NotImplementedError("output size != input size is not used on this path (train_mp3d.py:487-490)")
        D, dev = self.roi_heads.topk, self.device
        # this frame's own result tensors, from the caching allocator on the stream that fills them; `_materialize` slices them
        P["boxes"] = torch.empty((D, 4), dtype=torch.float32, device=dev)
        P["scores"] = torch.empty((D,), dtype=torch.float32, device=dev)
        P["classes"] = torch.empty((D,), dtype=torch.int32, device=dev)
        P["masks"] = torch.empty((D, H, W), dtype=torch.uint8, device=dev)
        ops.detector_postprocess(det_boxes, det_scores, det_classes, det_count, self.roi_heads.topk, out_w / W, out_h / H,
                                 float(out_w), float(out_h), P["boxes"], P["scores"], P["classes"], P["src"], P["count"],
                                 remap=self.roi_heads.last_selector.rep_of if self.dedup_detection_masks else None)
        ops.paste_masks(self.roi_heads.det_masks, P["boxes"], P["src"], P["count"], self.roi_heads.topk, out_h, out_w,
                        self.mask_threshold, P["masks"])

    def _enqueue_detection_pass(self, views, shapes, det, image_hw, frame, after=None):
        """`forward_with_given_boxes` (detic_roi_heads.py:257) + `detector_postprocess` + paste (custom_rcnn.py:579-580) of this
        frame on the detection stream (lowest priority: its GEMMs fill whatever the latency-bound chains of the frame -- and of
        the next frame -- leave idle)."""
        if self._det_stream is None:
            self._det_stream = _det_stream(self.device, self.det_stream_priority)
            self._ev_det = [torch.cuda.Event() for _ in range(RESULT_SETS)]
        ds = self._det_stream
        det_boxes, det_scores, det_classes, det_rows, det_count = det
        k = self._post_slot
        ds.wait_event(self._ev_box)
        if after is not None:
            ds.wait_event(after)
        with torch.cuda.stream(ds):
            self._mark("det_pass_begin", ds)
            self._detection_masks(views, shapes, det_boxes, det_count)
            self._postprocess_and_paste(det_boxes, det_scores, det_classes, det_count, image_hw, frame, self._post)
            self._ev_det[k].record(ds)
            self._mark("det_pass", ds)
        self._pyr_reader[self._pyramid] = self._ev_det[k]

    def select_memory_instances(self, prop_boxes, prop_scores, prop_count, image_hw):
        """`inference_with_proposals` up to the NMS (custom_rcnn.py:825-875): CLIP re-score of the proposals, threshold
        MEMORY_CLS_SCORE_THRESH, per-class NMS 0.5, top 100 -> (proposal row of every kept detection, count)."""
        H, W = image_hw
        R = self.proposal_generator.cap
        if self._mem_scores_frame != self._frame_no:        # not written by this frame's cascade (forward_box without mem_rescore)
            ops.memory_scores(self.roi_heads.featn0, self.zs_weight, prop_scores, self.mem_scores, prop_count, R, self.C1)
        _, _, _, rows, cnt = self.mem_selector(prop_boxes, self.mem_scores, prop_count, float(W), float(H), self.cls_score_thresh, 0.5)
        return rows, cnt

    def update_implicit_memory(self, prop_boxes, prop_scores, prop_count, prop_masks, proj, image_hw, mem_sel=None):
        rows, cnt = mem_sel if mem_sel is not None else self.select_memory_instances(prop_boxes, prop_scores, prop_count, image_hw)
        self._last_write = (prop_boxes, prop_masks, rows, cnt, proj)          # bench.py's HBM-class probe replays it
        follow = self.snapshot_follows_write if self.snapshot_follows_write is not None else self.test_type in ("default", "episodic")
        if follow and self._f16_valid and not self._dirty_pending:
            self._writer(self.roi_heads.featn0, prop_boxes, prop_masks, rows, cnt, proj, self.implicit_memory, self.observations,
                         err=self._err, snapshot=self._mem_f16)
        else:
            self._writer(self.roi_heads.featn0, prop_boxes, prop_masks, rows, cnt, proj, self.implicit_memory, self.observations,
                         dirty=self._dirty, err=self._err)
            self._dirty_pending = True

    def save_memory_snapshot(self, sequence_name: str) -> str:
        """`MODEL.TEST_SAVE_SEMMAP` dump (custom_rcnn.py:518-530): semmap, impicit_memory [sic], observations -> OUTPUT_DIR/memory/."""
        from ..data.snapshot import write_snapshot
        semmap = self.semantic_map()
        return write_snapshot(self.output_dir, sequence_name, semmap.cpu().numpy(), self.implicit_memory.cpu().numpy(),
                              self.observations.cpu().numpy())

    def _materialize(self, P) -> Instances:
        """Slice a result set by its detection count (the frame's only host wait: on the event recorded after the count's
        async copy).  The paste kernel writes 0/1 bytes, so the masks are handed out as a bool view's copy."""
        t0 = _time.perf_counter()
        P["ready"].synchronize()
        self.host_profile["wait_s"] += _time.perf_counter() - t0
        flags = int(P["err_host"][0])
        if flags:
            self._err.zero_()
            raise _lib.EodError(
                f"device error flags {flags:#x}: proj_indices holds cell indices outside [0, {self.implicit_memory.shape[0]}) "
                "(an index image written for another map size?); they were clamped, the frame's results are not trustworthy")
        n = int(P["count_host"][0])
        inst = Instances(P["hw"])
        cur = torch.cuda.current_stream(self.device)
        for k in ("boxes", "scores", "classes", "masks"):
            P[k].record_stream(cur)           # allocated on the stream that filled them; the caller reads them on this one
        inst.pred_boxes = Boxes(P["boxes"][:n])
        inst.scores = P["scores"][:n]
        inst.pred_classes = P["classes"][:n].to(torch.int64)
        inst.pred_masks = P["masks"][:n].view(torch.bool)       # the paste kernel writes 0/1 bytes: a view, no copy
        P["boxes"] = P["scores"] = P["classes"] = P["masks"] = None
        return inst

    def semantic_map(self) -> torch.Tensor:
        """a20, evaluated lazily: the explicit map `self.semmap` the reference recomputes (and syncs to the host) every
        frame (custom_rcnn.py:756) but only consumes when MODEL.TEST_SAVE_SEMMAP dumps it (518-530)."""
        self.semmap = ops.semmap_labels(self.implicit_memory, self.observations, self.zs_weight, self.obs_score_thresh)
        return self.semmap

    # ---- introspection used by tests / bench -----------------------------------------------------------
    def proposals_snapshot(self):
        dec = self.proposal_generator._plans[next(iter(self.proposal_generator._plans))][3]
        n = int(dec.count.item())
        return dict(proposal_boxes=dec.boxes[:n].cpu(), scores=dec.scores[:n].cpu(),
                    feat=self.roi_heads.feat0.view(-1, 512)[:n].cpu(), featn=self.roi_heads.featn0[:n].cpu(),
                    pred_masks=self.roi_heads.prop_masks[:n].cpu())
