#!/usr/bin/env python
"""`model(data); trainer.optimizer_step()` for many iterations on one batch of frames (two sequences x three frames, 640x640, a
200x200 memory per frame: the path `do_train` drives -- shared trunk pass, input prefetch on the copy stream, multi-tensor gradient
sums): device memory allocated / reserved at the start and at the end, ms per frame.  A leak check, not a benchmark.

    python tools/train_loop_soak.py [iterations]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.modeling.training import Trainer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
H = W = 640
n_cells = 200 * 200
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5, "FP16", False])
sd = synthetic_state_dict(0)
model = build_model(cfg, sd)
trainer = Trainer(model, sd)
g = torch.Generator().manual_seed(0)


def frame(i):
    xy = torch.rand((12, 2), generator=g) * torch.tensor([W * 0.6, H * 0.6])
    wh = torch.rand((12, 2), generator=g) * torch.tensor([W * 0.3, H * 0.3]) + 8
    obs = torch.randint(0, 6, (n_cells,), generator=g).float()
    return {"image": torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8),
            "instances": {"gt_boxes": torch.cat([xy, xy + wh], dim=1), "gt_classes": torch.randint(0, 20, (12,), generator=g)},
            "memory": (torch.randn((n_cells, 512), generator=g) * obs.clamp(min=1)[:, None]).numpy(), "observations": obs.numpy(),
            "proj_indices": torch.randint(0, n_cells, (H, W, 1), generator=g).numpy(), "sequence_name": f"s{i}", "memory_reset": i == 0}


data = [[frame(0), frame(1), frame(2)], [frame(3), frame(4), frame(5)]]
model.train()
for _ in range(3):
    model(data)
    trainer.optimizer_step()
torch.cuda.synchronize()
a0, r0 = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
t0 = time.perf_counter()
first = last = None
for it in range(iters):
    losses = model(data)
    trainer.optimizer_step()
    if it == 0 or it == iters - 1:
        v = sum(float(x) for x in losses.values())
        first, last = (v if first is None else first), v
torch.cuda.synchronize()
dt = time.perf_counter() - t0
a1, r1 = torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)
print(f"{iters} iterations of 6 frames: {dt / iters / 6 * 1e3:.2f} ms per frame; total loss {first:.3f} -> {last:.3f}; "
      f"allocated {a0 / 2**20:.0f} -> {a1 / 2**20:.0f} MB, reserved {r0 / 2**20:.0f} -> {r1 / 2**20:.0f} MB")
