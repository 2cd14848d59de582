#!/usr/bin/env python
"""One-stream time of the memory-independent trunk (ResNet-50 + FPN top-down) for N = 1, 2, 4 images of one size, planned like one
image (the N = B pass of BatchedSequences / the pair look-ahead): 10 back-to-back passes between one pair of events, / 10."""
import os, sys
import numpy as np
import torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_object_detection_amd import build_model, ops, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 640)
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5])
m = build_model(cfg, synthetic_state_dict(0))
img = (torch.rand((3, H, W)) * 255).to(torch.uint8).to(dev)
x4, Hp, Wp = ops.preprocess_image(img, m.pixel_mean, m.pixel_std)
for N in (1, 2, 4):
    x = torch.cat([x4] * N, dim=0)
    def run():
        c = m.backbone.bottom_up.forward(x, Hp, Wp, N=N) if N > 1 else m.backbone.bottom_up.forward(x, Hp, Wp)
        if N > 1:
            m.backbone.top_down_batched(c, Hp, Wp, N)
        else:
            m.backbone.top_down(c, Hp, Wp, 0)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _k in range(10):
            run()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    print(f"{H}x{W} trunk + top-down, N = {N}: {np.median(ts):6.3f} ms per pass, {np.median(ts) / N:6.3f} ms per image")
