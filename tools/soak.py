#!/usr/bin/env python
"""Longer bitwise comparison of the schedules than the test suite runs: N frames in episodes of 20 through
  (a) one stream, no look-ahead          (b) the default five-stream pipeline          (c) BatchedSequences of 2 (scene + a second scene)
  (d) LockstepScenes of 2 (N = 2 through every stage)
Detections, masks and the memory state of the scene must be identical in all four.

    python tools/soak.py [H W frames]"""
import os
import sys

import torch

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence
from embodied_object_detection_amd.modeling.batched import BatchedSequences
from embodied_object_detection_amd.modeling.lockstep import LockstepScenes

H, W, N = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 160, 100)
grid, cell = (200, 0.2) if H >= 480 else (24, 0.5)
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5])
sd = synthetic_state_dict(0)
dev = torch.device("cuda:0")


def frames_of(seq_id):
    seq = SyntheticSequence(seq_id, H=H, W=W, n_frames=N, map_w=grid, map_h=grid, cell=cell)
    out = []
    for i in range(N):
        f = seq.frame(i)
        f["image"] = f["image"].to(dev)
        f["proj_indices"] = torch.from_numpy(f["proj_indices"][..., 0]).to(dev)
        out.append(f)
    return out


def episodes(fr):
    return [fr[i:i + 20] for i in range(0, len(fr), 20)]


def keep(outs):
    return [(o["instances"].pred_boxes.tensor.clone(), o["instances"].scores.clone(), o["instances"].pred_classes.clone(),
             o["instances"].pred_masks.clone()) for o in outs]


fa, fb = frames_of(3), frames_of(4)
a = build_model(cfg, sd)
a.overlap_branches = False
a.prefetch_trunk = False
ra = []
for ep in episodes(fa):
    ra += keep(a([ep]))
b = build_model(cfg, sd)
rb = []
for ep in episodes(fa):
    rb += keep(b([ep]))
c = BatchedSequences(cfg, 2, sd)
rc = []
for ea, eb in zip(episodes(fa), episodes(fb)):
    rc += keep(c([ea, eb])[0])
d = LockstepScenes(cfg, 2, sd)
rd = []
for ea, eb in zip(episodes(fa), episodes(fb)):
    rd += keep(d([ea, eb])[0])


class _D:        # scene 0 of the lock-step batch with the single-scene model's attribute names
    implicit_memory, observations, _mem_f16 = d.implicit_memory[0], d.observations[0], d._mem_f16[0]


bad = 0
for name, other, m in (("pipeline", rb, b), ("lock-step (streams)", rc, c.scenes[0]), ("lock-step (launches)", rd, _D)):
    for i, (x, y) in enumerate(zip(ra, other)):
        if not all(torch.equal(p, q) for p, q in zip(x, y)):
            print(f"{name}: frame {i} differs")
            bad += 1
    if not (torch.equal(a.implicit_memory, m.implicit_memory) and torch.equal(a.observations, m.observations)):
        print(f"{name}: memory state differs")
        bad += 1
    if not torch.equal(a._mem_f16, m._mem_f16):
        print(f"{name}: fp16 snapshot differs")
        bad += 1
dets = sum(len(x[1]) for x in ra)
print(f"{H}x{W}, {N} frames, {dets} detections: {'IDENTICAL in all four schedules' if bad == 0 else f'{bad} MISMATCHES'}")
sys.exit(1 if bad else 0)
