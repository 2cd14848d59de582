"""Optimizer set-up of the reference's training configuration, host side (second slice).

Mirrors `build_custom_optimizer` (`Detic/detic/custom_solver.py:19-79`): one parameter group per trainable parameter, learning rate
`SOLVER.BASE_LR`, x `SOLVER.BACKBONE_MULTIPLIER` when "backbone" is in the parameter's name, x `SOLVER.CUSTOM_MULTIPLIER` when one
of `SOLVER.CUSTOM_MULTIPLIER_NAME` is (the recurrent configuration trains the `map_merge` projections at 10 x the base rate:
`configs/Detic_..._mp3d_recurrent.yaml:37-38`); `weight_decay` per group unless the optimizer is ADAMW (there it is the optimizer's
default); frozen parameters and repeated registrations of one tensor are skipped.  The learning-rate schedule is detectron2's
`WarmupCosineLR` (`Base-C2_L_R5021k_640b64_4x_recurrent.yaml:64-67`), restated from its published formula.  The update itself runs
on the device: `ops.AdamW` -> `eod_adamw_step`.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple


def match_name_keywords(name: str, keywords: Sequence[str]) -> bool:
    return any(k in name for k in keywords)


def build_param_groups(named_parameters: Iterable[Tuple[str, object]], base_lr: float, weight_decay: float, optimizer: str = "ADAMW",
                       backbone_multiplier: float = 1.0, custom_multiplier: float = 1.0,
                       custom_multiplier_name: Sequence[str] = (), frozen: Sequence[str] = ()) -> List[Dict]:
    """-> [{"name", "param", "lr"[, "weight_decay"]}, ...] in registration order (custom_solver.py:27-45).  `named_parameters`:
    (name, tensor) pairs; a parameter with `requires_grad == False` or a name in `frozen` is skipped, a tensor seen before (same
    object) is skipped.  The product keeps its parameters as plain device tensors (whose `requires_grad` is False by construction):
    for those only `frozen` decides."""
    if optimizer not in ("SGD", "ADAMW"):
        raise NotImplementedError(f"no optimizer type {optimizer}")          # custom_solver.py:76
    import torch
    groups, seen = [], set()
    for name, value in named_parameters:
        plain = type(value) is torch.Tensor
        if name in frozen or (not plain and not getattr(value, "requires_grad", True)):
            continue
        if id(value) in seen:
            continue
        seen.add(id(value))
        lr = base_lr
        if "backbone" in name:
            lr = lr * backbone_multiplier
        if match_name_keywords(name, custom_multiplier_name):
            lr = lr * custom_multiplier
        g = {"name": name, "param": value, "lr": lr}
        if optimizer != "ADAMW":
            g["weight_decay"] = weight_decay
        groups.append(g)
    return groups


def param_groups_from_cfg(cfg, named_parameters) -> List[Dict]:
    s = cfg.SOLVER
    return build_param_groups(named_parameters, float(s.BASE_LR), float(s.WEIGHT_DECAY), str(s.OPTIMIZER), float(s.BACKBONE_MULTIPLIER),
                              float(s.CUSTOM_MULTIPLIER), list(s.CUSTOM_MULTIPLIER_NAME))


def warmup_cosine_lr_factor(it: int, max_iter: int, warmup_iters: int, warmup_factor: float, warmup_method: str = "linear") -> float:
    """detectron2 `build_lr_scheduler` for LR_SCHEDULER_NAME WarmupCosineLR (called at train_mp3d.py:519): `LRMultiplier` over
    `WarmupParamScheduler(CosineParamScheduler(1, 0), warmup_factor, min(warmup_iters / max_iter, 1), method)`: the factor on every
    group's base rate when `it` scheduler steps have been taken.  Past the warmup it is the cosine at it / max_iter; inside, a line
    from warmup_factor x cosine(0) to the cosine's value where the warmup ends ("linear") or warmup_factor x cosine(0) ("constant")."""
    cosine = lambda where: 0.5 * (1.0 + math.cos(math.pi * where))
    where = it / max_iter
    wlen = min(warmup_iters / max_iter, 1.0)
    if where >= wlen:
        return cosine(min(where, 1.0))
    start = warmup_factor * cosine(0.0)
    if warmup_method == "constant":
        return start
    if warmup_method == "linear":
        a = where / wlen
        return cosine(wlen) * a + start * (1.0 - a)
    raise ValueError(f"Unknown warmup method: {warmup_method}")
