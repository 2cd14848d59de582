#!/usr/bin/env python
"""Time of one training iteration (`modeling/training.py`): by default the proposal half (`ProposalTrainer.step`: forward with kept
activations, target assignment, losses, backward through head / FPN / memory fusion / trunk, 96 AdamW launches, re-folding); with
`--roi-heads` the whole `forward_model` (`Trainer.step`: also train-mode proposals, matching / sampling, the cascade's three stages
with their losses and backward, 126 AdamW launches) -- on one synthetic 640x640 frame with 24 ground-truth boxes.  Diagnostics for
the training slices, not the headline metric.

    python tools/train_step_bench.py [--roi-heads] [--size 640 640] [--steps 10] [--warmup 3]
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o train -- python3 tools/train_step_bench.py
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.modeling.training import ProposalTrainer, Trainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=2, default=[640, 640])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--roi-heads", action="store_true", help="the whole forward_model: both halves, all 126 parameter tensors")
    ap.add_argument("--no-scale-sync", action="store_true",
                    help="experiment: skip the per-step read-back of the five Scale parameters (the one host synchronisation of a step)")
    ap.add_argument("--freeze-backbone", action="store_true",
                    help="MODEL.FREEZE_BACKBONE True with the shipped yaml's UNFROZEN_LAYERS ['roi', 'map_merge', 'proposal_generator']")
    a = ap.parse_args()
    H, W = a.size
    dev = torch.device("cuda:0")
    opts = ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5, "FP16", False]
    if a.freeze_backbone:
        opts += ["MODEL.FREEZE_BACKBONE", True, "MODEL.UNFROZEN_LAYERS", ["roi", "map_merge", "proposal_generator"]]
    cfg = setup_cfg(None, opts)
    sd = synthetic_state_dict(0)
    model = build_model(cfg, sd)
    trainer = Trainer(model, sd) if a.roi_heads else ProposalTrainer(model, sd)
    if a.no_scale_sync:
        trainer.after = [f for f in trainer.after if getattr(f, "__name__", "") != "sync_scales"]
    g = torch.Generator().manual_seed(0)
    n_cells = 200 * 200
    img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8).to(dev)
    mem16 = (torch.randn((n_cells, 512), generator=g) * 2).half().to(dev)
    proj = torch.randint(0, n_cells, (H, W), generator=g).int().to(dev)
    xy = torch.rand((24, 2), generator=g) * torch.tensor([W * 0.6, H * 0.6])
    wh = torch.rand((24, 2), generator=g) * torch.tensor([W * 0.35, H * 0.35]) + 8
    gt = torch.cat([xy, xy + wh], dim=1).to(dev)
    kw = dict(gt_classes=torch.randint(0, 20, (24,), generator=g).int().to(dev), generator=torch.Generator(device=dev).manual_seed(0)) \
        if a.roi_heads else {}
    losses = []
    for _ in range(a.warmup):
        losses.append(sum(float(v) for v in trainer.step(img, gt, memory=(mem16, proj), **kw).values()))
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = trainer.step(img, gt, memory=(mem16, proj), **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    losses.append(sum(float(v) for v in out.values()))
    extra = {"proposals": int(trainer.fm.last_proposals.shape[0]), "roi_rows_per_stage": [int(r["boxes"].shape[0]) for r in trainer.fm.det.last],
             "proposal_caps": [trainer.fm.pre, trainer.fm.post]} if a.roi_heads else {}
    print(json.dumps({"metric": "training_iterations_per_second" + ("" if a.roi_heads else "_proposal_half"), **extra, "value": round(1.0 / dt, 3), "ms_per_step": round(dt * 1e3, 2),
                      "size": [H, W], "gt_boxes": 24, "steps": a.steps, "warmup": a.warmup, "dtype": "f32",
                      "total_loss_first_last": [round(losses[0], 4), round(losses[-1], 4)],
                      "device_mb_allocated_before_after_the_timed_steps": [round(mem0 / 2**20, 1), round(torch.cuda.memory_allocated(dev) / 2**20, 1)],
                      "note": "one frame per iteration, parameters stepped in the layers the inference path runs"}))


if __name__ == "__main__":
    main()
