#!/usr/bin/env python
"""Which Python lines of a training iteration issue device copies / small torch kernels: `torch.profiler` with stacks over three
iterations of `Trainer.step` (640x640, 24 boxes), grouped by (op, innermost repo frame).  Diagnostics for the launch count of the
training step (`profiles/r04_train_step_*`), not a benchmark.

    python tools/train_copy_sites.py > gpurun_out/train_copy_sites.txt
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.modeling.training import Trainer

H = W = 640
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5, "FP16", False])
sd = synthetic_state_dict(0)
trainer = Trainer(build_model(cfg, sd), sd)
g = torch.Generator().manual_seed(0)
n_cells = 200 * 200
img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).to(dev)
mem16 = (torch.randn((n_cells, 512), generator=g) * 2).half().to(dev)
proj = torch.randint(0, n_cells, (H, W), generator=g).int().to(dev)
xy = torch.rand((24, 2), generator=g) * torch.tensor([W * 0.6, H * 0.6])
wh = torch.rand((24, 2), generator=g) * torch.tensor([W * 0.35, H * 0.35]) + 8
gt = torch.cat([xy, xy + wh], dim=1).to(dev)
kw = dict(gt_classes=torch.randint(0, 20, (24,), generator=g).int().to(dev), generator=torch.Generator(device=dev).manual_seed(0))
for _ in range(3):
    trainer.step(img, gt, memory=(mem16, proj), **kw)
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    for _ in range(STEPS):
        trainer.step(img, gt, memory=(mem16, proj), **kw)
    torch.cuda.synchronize()
sites = collections.Counter()
WATCH = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::zero_", "aten::cat", "aten::clone", "aten::sum",
         "aten::index", "aten::_to_copy", "aten::item", "aten::_local_scalar_dense")
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    where = "?"
    for fr in ev.stack:
        if "embodied_object_detection_amd" in fr or "tools/" in fr:
            where = fr.split("embodied_object_detection_amd/")[-1]
            break
    sites[(ev.name, where)] += 1
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n / STEPS:7.1f} per step  {name:28s} {where}")
