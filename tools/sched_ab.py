#!/usr/bin/env python
"""A/B of the frame schedule knobs in one process (GPU box): look-ahead enqueue order x arithmetic, models rebuilt each time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import build_model, ops, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5, "MODEL.DEVICE", "cuda:0"])
sd = synthetic_state_dict(0)
N = 46
seq = SyntheticSequence(0, H=640, W=640, n_frames=N, map_w=200, map_h=200, cell=0.2)
frames = []
for i in range(N):
    f = seq.frame(i); f["image"] = f["image"].to(dev); f["proj_indices"] = torch.from_numpy(f["proj_indices"][..., 0]).to(dev); frames.append(f)
def run(model):
    def step(i):
        if frames[i]["memory_reset"]: model.reset_memory(seq.n_cells)
        model.inference_frame(frames[i], materialize=False, next_frame=frames[i + 1] if i + 1 < N else None)
    for i in range(5): step(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(5, N - 1): step(i)
    torch.cuda.synchronize(); return (N - 6) / (time.perf_counter() - t)
for rep in range(2):
    for math in ("fp32", "bf16x3"):
        ops.set_conv_math(math)
        for first in (True, False):
            m = build_model(cfg, sd); m.lookahead_at_start = first
            print(f"{math:7s} lookahead_at_start={first!s:5s}  {run(m):7.1f} frames/s", flush=True)
            del m
ops.set_conv_math("fp32")
