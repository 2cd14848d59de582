#!/usr/bin/env python
"""Nothing but frames: 3 episodes of 20 resident 640x640 frames through the boundary (the first is warm-up), for a rocprofv3
`--kernel-trace --stats` run whose per-kernel call counts divide by the frame count without bench.py's probes, replays and variants.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pure -o pure -- python3 tools/pure_frames.py
    python tools/kernel_stats.py gpurun_out/prof_pure/pure_results.db profiles/r04_pure_frames_kernel_stats.csv
"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence

dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5,
                       "MODEL.MEMORY_CLS_SCORE_THRESH", 0.3, "MODEL.DEVICE", "cuda:0"])
model = build_model(cfg, synthetic_state_dict(0))
seq = SyntheticSequence(0, H=640, W=640, n_frames=60, map_w=200, map_h=200, cell=0.2)
frames = []
for i in range(60):
    f = seq.frame(i)
    f["image"] = f["image"].to(dev)
    f["proj_indices"] = torch.from_numpy(f["proj_indices"][..., 0]).to(dev)
    frames.append(f)
torch.cuda.synchronize()
model([frames[:20]])
torch.cuda.synchronize()
t = time.perf_counter()
n = len(model([frames[20:40]])) + len(model([frames[40:60]]))
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"{n} frames in {dt * 1e3:.1f} ms = {n / dt:.1f} frames/s (60 frames traced in total, 20 of them warm-up)")
