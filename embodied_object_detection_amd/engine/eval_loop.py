"""Eval driver of the hot path: episode loop, every-5th-frame evaluation, quartile bucketing, scene sharding and the
single-collective aggregation.

Mirrors `mp3d_inference_on_dataset` (`Detic/train_mp3d.py:85-363`) and `do_test` (`380-450`): per episode `model([inputs])`,
keep every 5th frame's output (`187-188`), rebuild the GT as integer-truncated XYWH boxes with `area: 0` (`232-238`),
bucket images into quartiles by `idx % 100` (`210-217`), evaluate the four quartiles and the whole set (`301-358`).

Multi-GPU: scenes are sharded `scene -> rank = scene_index % world`, all episodes of a scene stay on one rank in
order (the memory is per scene, `Detic/SMNet/loader.py:289-291`).  detectron2's `InferenceSampler` contiguous split is NOT used:
it would cut scenes across ranks and hit unset memory state (`custom_rcnn.py:485`).  Results are combined with ONE
`all_reduce(SUM)` of a fixed-shape record buffer (evaluation/coco_ap.py) instead of detectron2's pickle gather.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from ..evaluation.coco_ap import KIND_DET, KIND_GT, coco_eval

ROW = 10  # kind, scene id, frame index in scene, class, score, x1, y1, x2, y2, global episode index


def shard_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """Scene ids owned by `rank` (round-robin by scene id)."""
    return [s for s in range(n_scenes) if s % world == rank]


def gt_to_coco_xyxy(gt_boxes: torch.Tensor) -> np.ndarray:
    """The driver's GT: [int(x1), int(y1), int(x2-x1), int(y2-y1)] XYWH (train_mp3d.py:237) back to xyxy."""
    out = []
    for b in gt_boxes.tolist():
        x, y, w, h = int(b[0]), int(b[1]), int(b[2] - b[0]), int(b[3] - b[1])
        out.append([x, y, x + w, y + h])
    return np.asarray(out, dtype=np.float64).reshape(-1, 4)


class RecordBufferOverflow(RuntimeError):
    pass


def rows_needed(episodes, every: int = 5, max_dets: int = 100, max_gt: int = 156) -> int:
    """Capacity of the fixed-shape collective buffer for a rank.  `episodes`: the lengths of the episodes the rank runs (or one
    frame count, taken as ONE episode): every episode contributes ceil(len / every) evaluated frames (train_mp3d.py:187-188 takes
    frames 0, 5, 10, ... of EACH episode), each at most `max_dets` detections (COCO maxDets) and its GT boxes.  Every rank must
    pass the SAME number (the buffer shape is part of the collective), so callers take the maximum over ranks."""
    if isinstance(episodes, (int, np.integer)):
        episodes = [int(episodes)]
    imgs = sum((int(n) + every - 1) // every for n in episodes) + 1
    return max(1024, imgs * (max_dets + max_gt))


class RecordBuffer:
    """Host-side staging of this rank's records; `to_tensor` pads to the fixed [rows + 1, ROW] shape of the collective: the last
    row is a status row (kind 0, column 1 = number of records that did not fit).  A row that does not fit is COUNTED, not raised:
    raising on the overflowing rank alone would leave the other ranks waiting in the collective.  `gather_records` raises
    `RecordBufferOverflow` on EVERY rank after the all-reduce -- a silently truncated buffer would give a wrong AP."""

    def __init__(self, max_rows: int):
        self.max_rows = max_rows
        self.rows: List[List[float]] = []
        self.dropped = 0

    def add(self, row: Sequence[float]):
        if len(self.rows) >= self.max_rows:
            self.dropped += 1
            return
        self.rows.append(list(row))

    def to_tensor(self, device) -> torch.Tensor:
        t = torch.zeros((self.max_rows + 1, ROW), dtype=torch.float32)
        if self.rows:
            t[:len(self.rows)] = torch.tensor(self.rows, dtype=torch.float32)
        t[self.max_rows, 1] = float(self.dropped)
        return t.to(device)


def episode_offsets(frames_per_scene: Sequence[int], episode_len: int = 20) -> List[int]:
    """Global dataloader index of the first episode of every scene (dataset order = scene order, loader.py:97-105)."""
    off, run = [], 0
    for n in frames_per_scene:
        off.append(run)
        run += (n + episode_len - 1) // episode_len
    return off


def _record_episode(rec: "RecordBuffer", sid: int, im_id: int, idx: int, inputs, outputs, every: int) -> int:
    """Detections (COCO maxDets 100) and ground truth of every `every`-th frame of one episode (train_mp3d.py:187-188)."""
    outs = [outputs[i] for i in range(0, len(outputs), every)]
    ins = [inputs[i] for i in range(0, len(inputs), every)]
    for inp, out in zip(ins, outs):
        inst = out["instances"]
        n = min(len(inst), 100)                                     # COCO maxDets
        if n:
            b = inst.pred_boxes.tensor[:n].float().cpu().numpy()
            s = inst.scores[:n].float().cpu().numpy()
            c = inst.pred_classes[:n].cpu().numpy()
            for j in range(n):
                rec.add([KIND_DET, sid, im_id, float(c[j]), float(s[j]), *b[j].tolist(), idx])
        gt = inp.get("instances")
        if gt is not None:
            if isinstance(gt, dict):                                   # synthetic frames
                gboxes, gcls = gt["gt_boxes"], gt["gt_classes"]
            else:                                                       # Instances (map_mp3d_batch_to_coco)
                gboxes, gcls = gt.gt_boxes.tensor, gt.gt_classes
            gb = gt_to_coco_xyxy(gboxes)
            gc = gcls.tolist()
            for j in range(len(gc)):
                rec.add([KIND_GT, sid, im_id, float(gc[j]), 0.0, *gb[j].tolist(), idx])
        im_id += 1
    return im_id


def inference_on_scenes(model, scenes: Iterable, rank: int = 0, max_rows: int = 1 << 16, every: int = 5,
                        on_episode: Optional[Callable] = None, scene_episode_offset: Optional[Dict[int, int]] = None) -> Dict:
    """Run this rank's scenes; returns {'records': RecordBuffer, 'frames': n, 'seconds': t}.

    `scenes`: iterable of objects with `.seq_id` and `.episodes()` yielding lists of frame dicts (data/synthetic.py schema).
    Records carry (scene id, frame index) and the GLOBAL episode index, so the aggregate is identical however the scenes are
    sharded (COCO's stable sorts break score ties by image order).

    `model` is either the meta-architecture (`model([episode])`, one scene after the other, train_mp3d.py:186) or a
    `modeling.batched.BatchedSequences` of B scenes: this rank's scenes then run B at a time in lock-step (BASELINE configs[4];
    scenes are independent, so the records are the same set)."""
    rec = RecordBuffer(max_rows)
    frames = 0
    t0 = time.perf_counter()
    lockstep = len(getattr(model, "scenes", ())) if hasattr(model, "trunk_lookahead") else 0
    if lockstep > 1:
        scenes = list(scenes)
        for g in range(0, len(scenes), lockstep):
            group = scenes[g:g + lockstep]
            its = [iter(sc.episodes()) for sc in group]
            sids = [int(sc.seq_id) for sc in group]
            idxs = [(scene_episode_offset or {}).get(sid, 0) for sid in sids]
            im_ids = [0] * len(group)
            while True:
                eps = [next(it, None) for it in its]
                if all(e is None for e in eps):
                    break
                outs = model(eps + [None] * (lockstep - len(group)))
                for k, e in enumerate(eps):
                    if e is None:
                        continue
                    frames += len(e)
                    im_ids[k] = _record_episode(rec, sids[k], im_ids[k], idxs[k], e, outs[k], every)
                    if on_episode is not None:
                        on_episode(idxs[k], e, outs[k])
                    idxs[k] += 1
    else:
        for scene in scenes:
            sid = int(scene.seq_id)
            idx = (scene_episode_offset or {}).get(sid, 0)
            im_id = 0
            for inputs in scene.episodes():
                outputs = model([inputs])                                      # train_mp3d.py:186
                frames += len(inputs)
                im_id = _record_episode(rec, sid, im_id, idx, inputs, outputs, every)
                if on_episode is not None:
                    on_episode(idx, inputs, outputs)
                idx += 1
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return {"records": rec, "frames": frames, "seconds": time.perf_counter() - t0}


def gather_records(rec: RecordBuffer, rank: int, world: int, device) -> np.ndarray:
    """ONE collective: every rank writes its slice of a zero [world, rows + 1, ROW] buffer, all_reduce(SUM).  The status rows
    travel with it, so an overflow on any rank raises on all of them, after the collective."""
    local = rec.to_tensor(device)
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        if world != 1:
            raise RuntimeError(f"world={world} but torch.distributed is not initialised")
        out = local.cpu().numpy()[None]
    else:
        # also with ONE rank the buffer goes through the collective (RCCL on GPUs, gloo on CPU): same code path at every size
        buf = torch.zeros((world,) + tuple(local.shape), dtype=torch.float32, device=device)
        buf[rank] = local
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        out = buf.cpu().numpy()
    dropped = out[:, -1, 1]
    if dropped.sum() > 0:
        raise RecordBufferOverflow(
            "record buffer full: " + ", ".join(f"rank {r} dropped {int(d)} records" for r, d in enumerate(dropped) if d > 0)
            + f" ({rec.max_rows} rows per rank): size it with eval_loop.rows_needed(episode lengths of the busiest rank) -- the "
              "aggregate AP would silently lose detections otherwise")
    return out[:, :-1]


def records_by_image(buf: np.ndarray):
    """Gathered [world, rows, ROW] records -> (detections, ground truth, quartile) keyed by the canonical image id
    (scene * 1e6 + evaluated-frame index in the scene).  Quartile = the driver's `idx % 100` buckets of the GLOBAL episode index
    (train_mp3d.py:210-217)."""
    dets, gts, quart = {}, {}, {}
    for r in range(buf.shape[0]):
        rows = buf[r]
        rows = rows[rows[:, 0] > 0]
        for row in rows:
            uid = int(row[1]) * 1_000_000 + int(row[2])                   # canonical image order: (scene, frame)
            store = dets if row[0] == KIND_DET else gts
            e = store.setdefault(uid, {"boxes": [], "scores": [], "classes": []})
            e["boxes"].append(row[5:9])
            e["scores"].append(row[4])
            e["classes"].append(int(row[3]))
            quart[uid] = min(3, int(row[9]) % 100 // 25)                  # idx % 100 buckets (:210-217)
    for store in (dets, gts):
        for e in store.values():
            e["boxes"] = np.asarray(e["boxes"], dtype=np.float64).reshape(-1, 4)
            e["scores"] = np.asarray(e["scores"], dtype=np.float64)
            e["classes"] = np.asarray(e["classes"], dtype=np.int64)
    return dets, gts, quart


def evaluate_gathered(buf: np.ndarray, num_classes: int) -> Dict[str, Dict[str, float]]:
    """Quartile + overall COCO bbox results (train_mp3d.py:301-358) from the gathered [world, rows, ROW] records."""
    dets, gts, quart = records_by_image(buf)
    all_ids = sorted(set(dets) | set(gts))
    results = {"all": coco_eval(dets, gts, num_classes, image_ids=all_ids)}
    for qi, name in enumerate(("first_quartile", "second_quartile", "third_quartile", "fourth_quartile")):
        ids = [u for u in all_ids if quart.get(u) == qi]
        if ids:
            results[name] = coco_eval(dets, gts, num_classes, image_ids=ids)
    return results
