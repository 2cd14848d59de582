#!/usr/bin/env python
"""Golden vectors of the optimizer set-up  --  runs ONLY in the development container (needs /root/reference).

Executes the reference's OWN `build_custom_optimizer` (`Detic/detic/custom_solver.py:19-79`) on a small module whose parameter
names follow the model's (`backbone.map_merge_projection1.weight`, `backbone.bottom_up...`, `roi_heads...`, a frozen parameter, one
tensor registered under two names) for three solver configurations, and stores what it builds: the parameter groups in order
(name, lr, weight_decay or absent), the optimizer class, its defaults, and -- one step of that optimizer on seeded gradients --
the updated parameters (the reference's update rule as torch executes it).

Recipe as in gen_golden.py: detectron2 is absent and answered by stubs; the one detectron2 function the module
calls on this path, `maybe_add_gradient_clipping(cfg, optimizer)`, is restated from its published semantics for CLIP_TYPE "value"
(clip every parameter's gradient to [-CLIP_VALUE, CLIP_VALUE] before the step).  No reference source or bytecode is copied.

    python tests/golden/gen_golden_solver.py        # writes tests/golden/solver.json
"""
from __future__ import annotations

import json
import os
import sys
import types

import torch
import torch.nn as nn

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402   (the shim machinery)

NAMES = ["backbone.bottom_up.base.conv1.weight", "backbone.map_merge_projection1.weight", "backbone.map_merge_projection1.bias",
         "backbone.fpn_lateral3.weight", "proposal_generator.centernet_head.bbox_tower.0.weight", "roi_heads.box_head.0.fc1.weight",
         "roi_heads.mask_head.map_merge_like.weight"]
FROZEN = {"backbone.fpn_lateral3.weight"}
SHAPES = [(4, 3), (6, 5), (6,), (3, 3), (5, 2), (7, 4), (2, 2)]


class Tiny(nn.Module):
    """named_parameters() yields NAMES in order, then the first tensor once more under another name (custom_solver.py:32-35)."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.ps = [nn.Parameter(torch.randn(s, generator=g)) for s in SHAPES]
        for n, p in zip(NAMES, self.ps):
            p.requires_grad = n not in FROZEN

    def named_parameters(self, recurse=True):
        for n, p in zip(NAMES, self.ps):
            yield n, p
        yield "alias.of.conv1.weight", self.ps[0]


def ns(**k):
    return types.SimpleNamespace(**k)


def solver_cfg(optimizer, base_lr, wd, backbone_mult, custom_mult, names, clip_type):
    return ns(SOLVER=ns(CUSTOM_MULTIPLIER_NAME=names, OPTIMIZER=optimizer, BASE_LR=base_lr, WEIGHT_DECAY=wd, BACKBONE_MULTIPLIER=backbone_mult,
                        CUSTOM_MULTIPLIER=custom_mult, MOMENTUM=0.9, NESTEROV=False,
                        CLIP_GRADIENTS=ns(ENABLED=True, CLIP_TYPE=clip_type, CLIP_VALUE=1.0, NORM_TYPE=2.0)))


def clip_by_value(cfg, optimizer):
    """detectron2.solver.build.maybe_add_gradient_clipping, CLIP_TYPE "value": the returned optimizer clips each parameter's gradient
    to +-CLIP_VALUE before every step.  Here: the instance's step is wrapped."""
    if not cfg.SOLVER.CLIP_GRADIENTS.ENABLED:
        return optimizer
    inner = optimizer.step

    def step(closure=None):
        for grp in optimizer.param_groups:
            for p in grp["params"]:
                if p.grad is not None:
                    p.grad.clamp_(-cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE, cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE)
        return inner(closure)

    optimizer.step = step
    return optimizer


def main():
    G.STUB_ROOTS = ("detectron2", "fvcore")
    G.install_shim()
    import detectron2.solver.build as d2b       # the stub module
    d2b.maybe_add_gradient_clipping = clip_by_value
    cs = G._load("detic_custom_solver", os.path.join(G.DETIC, "detic", "custom_solver.py"))
    cases = {
        "recurrent_yaml": solver_cfg("ADAMW", 1e-5, 1e-4, 1.0, 10.0, ["map_merge"], "value"),      # configs/..._mp3d_recurrent.yaml:28-38
        "sgd_backbone_multiplier": solver_cfg("SGD", 0.02, 1e-4, 0.1, 10.0, ["map_merge", "bbox_tower"], "value"),
        "adamw_full_model_clip": solver_cfg("ADAMW", 2e-4, 1e-4, 1.0, 1.0, [], "full_model"),
    }
    out = {"names": NAMES, "shapes": SHAPES, "frozen": sorted(FROZEN), "cases": {}}
    for key, cfg in cases.items():
        model = Tiny()
        opt = cs.build_custom_optimizer(cfg, model)
        name_of = {id(p): n for n, p in zip(NAMES, model.ps)}
        groups = []
        for grp in opt.param_groups:
            assert len(grp["params"]) == 1
            e = {"name": name_of[id(grp["params"][0])], "lr": grp["lr"], "weight_decay": grp["weight_decay"]}
            groups.append(e)
        # which groups carried an explicit weight_decay is visible from the value: AdamW's default (0.0001 here, passed to its
        # constructor) or the group's own
        g = torch.Generator().manual_seed(9)
        before = [p.detach().clone() for p in model.ps]
        grads = []
        for p in model.ps:
            gr = torch.randn(p.shape, generator=g) * 3.0          # some |g| > CLIP_VALUE
            grads.append(gr)
            if p.requires_grad:
                p.grad = gr.clone()
        opt.step()
        opt.step()                                                # same gradients again (clamped in place the first time)
        out["cases"][key] = {
            "solver": {k: (v if not isinstance(v, types.SimpleNamespace) else vars(v)) for k, v in vars(cfg.SOLVER).items()},
            "optimizer_bases": [c.__name__ for c in type(opt).__mro__ if c.__module__.startswith("torch.optim")][:1],
            "defaults": {k: opt.defaults[k] for k in ("lr", "betas", "eps", "weight_decay", "momentum", "nesterov") if k in opt.defaults},
            "groups": groups,
            "params_before": [b.reshape(-1).tolist() for b in before],
            "grads": [gr.reshape(-1).tolist() for gr in grads],
            "params_after_two_steps": [p.detach().reshape(-1).tolist() for p in model.ps],
        }
    path = os.path.join(HERE, "solver.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print("wrote", path, {k: len(v["groups"]) for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
