#!/usr/bin/env python
"""Per-frame schedule of the multi-stream frame from timing events recorded by the model itself (model.trace): when each stage
of frame t ends, relative to the start of frame t, median over the steady-state frames.  No profiler: under rocprofv3 the host
needs > 6 ms to enqueue a frame and the GPU timeline is host-bound, which it is not in a normal run."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence

H = W = 640
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5])
LOCKSTEP = int(os.environ.get("LOCKSTEP", 0))      # B > 1: the schedule of a LockstepScenes step (B frames) instead of a frame
if LOCKSTEP > 1:
    from embodied_object_detection_amd.modeling.lockstep import LockstepScenes
    model = LockstepScenes(cfg, LOCKSTEP, synthetic_state_dict(0))
else:
    model = build_model(cfg, synthetic_state_dict(0))
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    setattr(model, k, eval(v))


def resident(seed):
    seq = SyntheticSequence(seed, H=H, W=W, n_frames=60)
    out = []
    for i in range(60):
        f = seq.frame(i)
        f["image"] = f["image"].to(dev)
        f["proj_indices"] = torch.from_numpy(f["proj_indices"][..., 0]).to(dev)
        out.append(f)
    return out


import time
if LOCKSTEP > 1:
    eps = [resident(b) for b in range(LOCKSTEP)]
    model([e[:20] for e in eps])
    torch.cuda.synchronize()
    model.trace = []
    t0 = time.perf_counter()
    model([e[20:40] for e in eps])
    model([e[40:60] for e in eps])
    torch.cuda.synchronize()
    print(f"{40 * LOCKSTEP / (time.perf_counter() - t0):.1f} frames/s with the trace events in ({LOCKSTEP} scenes in lock-step: times per STEP)")
else:
    frames = resident(0)
    model([frames[:20]])
    torch.cuda.synchronize()
    model.trace = []
    t0 = time.perf_counter()
    model([frames[20:40]])
    model([frames[40:60]])
    torch.cuda.synchronize()
    print(f"{40 / (time.perf_counter() - t0):.1f} frames/s with the trace events in")
by = collections.defaultdict(dict)
for fr, name, ev in model.trace:
    by[fr][name] = ev
frs = sorted(by)
rel = collections.defaultdict(list)
for a, b in zip(frs[5:-3], frs[6:-2]):
    s0 = by[a]["start"]
    for name, ev in by[a].items():
        rel[name].append(s0.elapsed_time(ev))
    rel["next_frame_start"].append(s0.elapsed_time(by[b]["start"]))
for name, v in sorted(rel.items(), key=lambda kv: np.median(kv[1])):
    print(f"{name:24s} {np.median(v):7.3f} ms   (p10 {np.percentile(v, 10):6.3f}, p90 {np.percentile(v, 90):6.3f})")
