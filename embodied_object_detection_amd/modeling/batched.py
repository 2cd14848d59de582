"""B independent sequences in lock-step on one GPU (BASELINE.json configs[4]: 4 sequences at 960x960, 512x512 memory grid).

Only independent sequences may be batched: inside a sequence frame t+1 reads the memory frame t wrote
(`Detic/detic/modeling/meta_arch/custom_rcnn.py:485-515`, `Detic/SMNet/loader.py:289-293`), so consecutive frames of one scene are
never put in a batch.

What is batched: the memory-independent half of the frame -- `preprocess_image`, the ResNet-50 trunk and the FPN top-down convs
(`backbone/timm.py:277-299,118-136`, ~75 launches, two thirds of a frame's launches) run ONCE per step for all B images (N = B
through `eod_conv2d`, planned like one image so that every image's result is bitwise the single-image result).  The scenes then
continue on their own streams with their own memory state: memory read + fusion, proposals, cascade, mask passes, memory write
(`CustomRCNNRecurrent.inference_frame` takes the batched trunk output exactly as it takes its own look-ahead).  The weights exist
once; every scene has its own activation buffers and memory.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import ops
from .meta_arch import PYRAMID_SETS, CustomRCNNRecurrent


def _share_weights(dst: CustomRCNNRecurrent, src: CustomRCNNRecurrent):
    """Point every conv / linear layer of `dst` at `src`'s device weights (the layer objects stay separate: each keeps its own
    launch descriptor)."""
    def convs(m: CustomRCNNRecurrent):
        out = [m.backbone.bottom_up.stem]
        for (_li, c1, c2, c3, ds) in m.backbone.bottom_up.blocks:
            out += [c1, c2, c3] + ([ds] if ds is not None else [])
        for l in (3, 4, 5):
            out += [m.backbone.lateral[l], m.backbone.output[l]]
        out += [m.backbone.p6, m.backbone.p7]
        out += [t[0] for t in m.proposal_generator.tower] + [m.proposal_generator.out_conv]
        for st in m.roi_heads.stages:
            out += [st["fc1"], st["fc2"], st["cls"], st["bb0"], st["bb2"]]
        out += list(m.roi_heads.mask_convs) + [m.roi_heads.deconv]
        return out

    for a, b in zip(convs(dst), convs(src)):
        a.w, a.bias, a.w_split = b.w, b.bias, b.w_split
    dst.backbone.merge.prepared = src.backbone.merge.prepared
    for a, b in zip(dst.proposal_generator.tower, src.proposal_generator.tower):
        pass   # GroupNorm gamma / beta are tiny; left per scene
    for sa, sb in zip(dst.roi_heads.stages, src.roi_heads.stages):
        sa["zs"] = sb["zs"]
    dst.roi_heads.pred_w = src.roi_heads.pred_w
    dst.zs_weight = src.zs_weight


class BatchedSequences:
    """`BatchedSequences(cfg, B)(episodes)`: `episodes` = list of B frame lists, one per sequence (`None` or `[]`: that sequence
    sits this call out; lengths may differ: a sequence whose episode has ended drops out of the lock-step, the trunk then runs
    with N = the number still active); returns a list of B output lists, each what `CustomRCNNRecurrent.forward([episode_b])`
    returns."""

    def __init__(self, cfg, batch: int, state_dict: Optional[Dict[str, torch.Tensor]] = None, concurrent_scenes: int = 0):
        """`concurrent_scenes`: how many scenes may be in flight at once (scene b runs on stream b % concurrent_scenes; 0 = the
        default, 2).  Two are enough to put one scene's latency-bound front beside the other's dense mask passes; more only evict
        each other's tiles from the L2s.  Measured, B = 4: 960x960 158.6 / 157.0 / 150.2 frames/s and 640x640 210.4 / 204.7 / 201.1
        frames/s with 2 / 3 / 4 in flight."""
        if batch < 1:
            raise ValueError("batch must be >= 1")
        self.concurrent_scenes = int(concurrent_scenes)
        self.scenes: List[CustomRCNNRecurrent] = [CustomRCNNRecurrent(cfg, state_dict) for _ in range(batch)]
        for m in self.scenes[1:]:
            _share_weights(m, self.scenes[0])
        torch.cuda.empty_cache()                 # the duplicate weight uploads of scenes 1.. are released
        self.device = self.scenes[0].device
        self._stream_pool = [torch.cuda.Stream(device=self.device, priority=-1) for _ in range(batch)]
        self.streams = list(self._stream_pool)
        self.trunk_stream = torch.cuda.Stream(device=self.device, priority=-1)
        self._ev_in = torch.cuda.Event()
        self._ev_trunk = torch.cuda.Event()
        self._ev_done = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(batch)]      # [scene][step parity]
        for m in self.scenes:
            # the trunk comes from the batched pass; inside a scene the box cascade / memory write still run beside the mask passes
            # (`intra_scene_overlap`), the scenes overlap each other on their own streams
            m.prefetch_trunk = False
        self.intra_scene_overlap = False     # measured at 960x960, B = 4: 121.6 frames/s in order vs 89.5 with the shared side streams
        # the batched trunk of step t + 1 depends on the images only: it runs on its own stream beside the scenes' chains of step t
        # (the batch's form of the single-sequence look-ahead) instead of as a barrier between two steps
        self.trunk_lookahead = True
        # scenes that start a call together stay in phase: their latency-bound fronts coincide and so do their dense mask
        # passes.  With the stagger, scene b starts its first frame of a call when scene b - 1 has finished the FRONT of its
        # first frame (memory fusion .. box cascade, ~40 % of a frame): the dense half of one scene runs beside the front of the next
        self.stagger = True
        for m in self.scenes:
            m.front_event = torch.cuda.Event()

    def __call__(self, episodes: List[List[dict]]):
        return self.forward(episodes)

    def _batched_trunk(self, scenes: List[CustomRCNNRecurrent], frames: List[dict]):
        """One N = len(frames) pass of preprocess + ResNet-50 + FPN top-down; every scene's P3..P5 are copied into the pyramid set
        its next frame reads, and the scene is told that its trunk has been computed ahead."""
        m0 = self.scenes[0]
        B = len(frames)
        H, W = int(frames[0]["image"].shape[-2]), int(frames[0]["image"].shape[-1])
        if any((int(f["image"].shape[-2]), int(f["image"].shape[-1])) != (H, W) for f in frames):
            raise ValueError("the frames of one lock-step must have one size")
        xs = []
        for m, f in zip(scenes, frames):
            x4, Hp, Wp = ops.preprocess_image(m._device_image(f), m.pixel_mean, m.pixel_std)
            xs.append(x4)
        x = torch.cat(xs, dim=0)                 # [B,Hp,Wp,4] (a device copy; no arithmetic)
        c = m0.backbone.bottom_up.forward(x, Hp, Wp, N=B)
        p345 = m0.backbone.top_down_batched(c, Hp, Wp, B)
        for b, (m, f) in enumerate(zip(scenes, frames)):
            nxt = (m._pyramid + 1) % PYRAMID_SETS
            shapes, off, feats, views, pooled = m.backbone._plan(Hp, Wp, nxt)
            for l in range(3):
                views[l].copy_(p345[l][b:b + 1])
            m._prefetched = (f["image"], Hp, Wp)
        return Hp, Wp

    def forward(self, episodes: List[Optional[List[dict]]]):
        B = len(self.scenes)
        if len(episodes) != B:
            raise ValueError(f"need {B} episodes (None for a sequence that sits this call out)")
        episodes = [e if e else [] for e in episodes]
        T = max(len(e) for e in episodes)
        first = next((e[0] for e in episodes if e), None)
        if first is not None:
            n_conc = self.concurrent_scenes if self.concurrent_scenes > 0 else 2
            self.streams = [self._stream_pool[b % max(1, min(n_conc, B))] for b in range(B)]
        outs: List[List[dict]] = [[] for _ in range(B)]
        pending: List[List] = [[] for _ in range(B)]
        cur = torch.cuda.current_stream(self.device)
        ts = self.trunk_stream if self.trunk_lookahead else cur
        active = lambda t: [b for b in range(B) if t < len(episodes[b])]

        def enqueue_trunk(t: int):
            """the batched trunk of step t on `ts`; it overwrites, for every active scene, the pyramid set of its frame t - PYRAMID_SETS"""
            act = active(t)
            for b in act:
                m = self.scenes[b]
                # with the look-ahead, frame t - 1 is only being enqueued now; what must be over is frame t - 2 (scenes run their
                # frames in order, so its event covers every earlier user of that set); without it, frame t - 1
                if t >= (2 if self.trunk_lookahead else 1):
                    ts.wait_event(self._ev_done[b][t % 2 if self.trunk_lookahead else (t - 1) % 2])
                rd = m._pyr_reader.get((m._pyramid + 1) % PYRAMID_SETS)
                if rd is not None:
                    ts.wait_event(rd)                    # a trailing detection pass may still read that set
            with torch.cuda.stream(ts):
                self._batched_trunk([self.scenes[b] for b in act], [episodes[b][t] for b in act])
                self._ev_trunk.record(ts)

        self._ev_in.record(cur)
        if self.trunk_lookahead:
            ts.wait_event(self._ev_in)
        if T:
            enqueue_trunk(0)
        for t in range(T):
            for b in active(t):
                m, f, s = self.scenes[b], episodes[b][t], self.streams[b]
                s.wait_event(self._ev_trunk)
                if t == 0:
                    s.wait_event(self._ev_in)
                    prev = [a for a in active(0) if a < b]
                    if self.stagger and prev:
                        s.wait_event(self.scenes[prev[-1]].front_event)
                with torch.cuda.stream(s):
                    if f["memory_reset"]:
                        m.reset_memory(int(episodes[b][0]["memory"].shape[0]))
                    if m.implicit_memory is None:
                        raise RuntimeError("first frame of a scene must carry memory_reset=True")
                    refresh = m.test_type in ("default", "episodic") or (m.test_type == "longterm" and t == 0)
                    m._ev_trunk = self._ev_trunk          # what `inference_frame` waits on before it takes a trunk computed ahead
                    m.overlap_branches = self.intra_scene_overlap
                    m.inference_frame(f, refresh_memory_snapshot=refresh, materialize=False)
                    pending[b].append(m._post_ticket())
                    self._ev_done[b][t % 2].record(s)
            if t + 1 < T:
                enqueue_trunk(t + 1)
            for b, m in enumerate(self.scenes):
                if len(pending[b]) == 3:
                    with torch.cuda.stream(self.streams[b]):
                        outs[b].append({"instances": m._materialize(pending[b].pop(0))})
        for b, m in enumerate(self.scenes):
            with torch.cuda.stream(self.streams[b]):
                for ticket in pending[b]:
                    outs[b].append({"instances": m._materialize(ticket)})
            if episodes[b]:
                cur.wait_event(self._ev_done[b][(len(episodes[b]) - 1) % 2])
        torch.cuda.current_stream(self.device).synchronize()
        return outs
