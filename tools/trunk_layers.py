#!/usr/bin/env python
"""Per-layer time of the memory-independent trunk pass (ResNet-50 + FPN top-down) alone on the chip, for N = 1 and N = 2 images (the
look-ahead's pair pass): HIP events around every `eod_conv2d` call (its slab reduce included), median of 5 passes.

    python tools/trunk_layers.py > gpurun_out/trunk_layers.txt
"""
import os
import sys

import numpy as np
import torch

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_object_detection_amd import build_model, ops, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict

H = W = 640
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5])
m = build_model(cfg, synthetic_state_dict(0))
img = (torch.rand((3, H, W)) * 255).to(torch.uint8).to(dev)
x4, Hp, Wp = ops.preprocess_image(img, m.pixel_mean, m.pixel_std)
orig = ops.Conv.__call__
meta = []


def call(self, x, N, Hh, Ww, **k):
    out = orig(self, x, N, Hh, Ww, **k)
    OH, OW = self.out_hw(Hh, Ww)
    meta.append((self.name, N * OH * OW, self.Cout, self.KH * self.KW * self.Cin))
    return out


for N in (1, 2):
    x = torch.cat([x4] * N, dim=0)

    def run():
        c = m.backbone.bottom_up.forward(x, Hp, Wp, N=N) if N > 1 else m.backbone.bottom_up.forward(x, Hp, Wp)
        if N > 1:
            m.backbone.top_down_batched(c, Hp, Wp, N)
        else:
            m.backbone.top_down(c, Hp, Wp, 0)
    run()
    torch.cuda.synchronize()
    convs = []

    def walk(o, seen):
        if id(o) in seen:
            return
        seen.add(id(o))
        if isinstance(o, ops.Conv):
            convs.append(o)
            return
        if isinstance(o, dict):
            [walk(v, seen) for v in o.values()]
        elif isinstance(o, (list, tuple)):
            [walk(v, seen) for v in o]
        elif hasattr(o, "__dict__"):
            [walk(v, seen) for v in vars(o).values()]
    walk(m.backbone, set())
    runs = []
    for rep in range(5):
        log = []
        for c in convs:
            c.event_log = log
        meta.clear()
        ops.Conv.__call__ = call
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        ops.Conv.__call__ = orig
        runs.append(([e0.elapsed_time(e1) * 1e3 for e0, e1, _ in log], a.elapsed_time(b) * 1e3, list(meta)))
    for c in convs:
        c.event_log = None
    us = np.median(np.array([r[0] for r in runs]), axis=0)
    tot = float(np.median([r[1] for r in runs]))
    print(f"N = {N}: pass {tot:.0f} us with the events in, conv calls {us.sum():.0f} us, {len(us)} calls")
    print(f"{'layer':48s} {'M':>7s} {'N':>5s} {'K':>5s} {'us':>7s} {'TFLOP/s':>8s}")
    for (name, M, Nn, K), t in zip(runs[0][2], us):
        print(f"{name[-48:]:48s} {M:7d} {Nn:5d} {K:5d} {t:7.1f} {2.0 * M * Nn * K / t / 1e6:8.1f}")
