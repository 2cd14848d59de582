"""`do_train` of the reference's driver, host side (`Detic/train_mp3d.py:509-659`) around the device step of `modeling/training.py`.

The reference's iteration -- `data = map_mp3d_batch_to_coco(data); loss_dict = model(data); losses.backward(); optimizer.step();
scheduler.step(); periodic_checkpointer.step(iteration)` -- with the same sampler semantics (detectron2's `TrainingSampler`: an
infinite stream of seeded shuffles of the episode indices, `IMS_PER_BATCH` episodes per iteration, `drop_last`), the same schedule
(`WarmupCosineLR` as `solver.warmup_cosine_lr_factor` restates it), the same finite-loss assertion (:612) and the same checkpoint
rhythm (`PeriodicCheckpointer`: every `CHECKPOINT_PERIOD` iterations and `model_final` at the end).  What is not here: the AMP
GradScaler (this path computes in fp32), DataLoader worker processes, TensorBoard / JSON writers (a `log` callable takes their rows).
"""
from __future__ import annotations

import math
import os
import time
from typing import Callable, Dict, Iterator, List, Optional

import torch


def training_sampler(size: int, seed: int = 0, shuffle: bool = True) -> Iterator[int]:
    """detectron2 `TrainingSampler._infinite_indices` for one rank: seeded permutations of range(size), one after another, forever."""
    g = torch.Generator()
    g.manual_seed(seed)
    while True:
        order = torch.randperm(size, generator=g).tolist() if shuffle else list(range(size))
        yield from order


def training_batches(dataset, ims_per_batch: int, seed: int = 0, shuffle: bool = True, collate: Optional[Callable] = None) -> Iterator[List]:
    """The DataLoader of `train_mp3d.py:563-572`: `ims_per_batch` episodes per iteration drawn by `training_sampler`, collated by
    `collate_smnet` (a list of episodes; an episode is the loader's list of frame records)."""
    if ims_per_batch < 1 or len(dataset) < 1:
        raise ValueError("training needs at least one episode per batch and a non-empty dataset")
    it = training_sampler(len(dataset), seed, shuffle)
    while True:
        batch = [dataset[next(it)] for _ in range(ims_per_batch)]
        yield collate(batch) if collate is not None else batch


def lr_factor_at(cfg, iteration: int, max_iter: int) -> float:
    from .. import solver
    s = cfg.SOLVER
    if str(s.LR_SCHEDULER_NAME) != "WarmupCosineLR":
        raise NotImplementedError("SOLVER.LR_SCHEDULER_NAME: WarmupCosineLR (Base-C2_L_R5021k_640b64_4x_recurrent.yaml:64)")
    return solver.warmup_cosine_lr_factor(iteration, max_iter, int(s.WARMUP_ITERS), float(s.WARMUP_FACTOR), str(s.WARMUP_METHOD))


def do_train(cfg, model, trainer, batches: Iterator[List], *, start_iter: int = 0, max_iter: Optional[int] = None, output_dir: Optional[str] = None,
             base_state_dict: Optional[Dict[str, torch.Tensor]] = None, map_batch: Optional[Callable] = None,
             log: Optional[Callable[[Dict], None]] = None, log_period: int = 20) -> List[Dict]:
    """-> one row per iteration {iteration, total_loss, <loss names>, lr, time, data_time}.  `batches`: iterator of loader batches
    (`training_batches`); `map_batch`: `map_mp3d_batch_to_coco` (None: the batches already hold frame dicts).  Checkpoints (when
    `output_dir` and `base_state_dict` are given): `model_{iteration:07d}.pth` every SOLVER.CHECKPOINT_PERIOD iterations, `model_final.pth`
    after the last one, in DetectionCheckpointer's format (`checkpoint.save_checkpoint`)."""
    from .. import checkpoint
    s = cfg.SOLVER
    if max_iter is None:
        max_iter = int(s.MAX_ITER) if int(s.get("TRAIN_ITER", -1)) < 0 else int(s.TRAIN_ITER)          # train_mp3d.py:529
    period = int(s.CHECKPOINT_PERIOD)
    base_lr = float(s.BASE_LR)
    model.train()
    rows: List[Dict] = []
    t_data = time.perf_counter()
    for data, iteration in zip(batches, range(start_iter, max_iter)):
        if map_batch is not None:
            data = map_batch(data)
        data_time = time.perf_counter() - t_data
        t0 = time.perf_counter()
        loss_dict = model(data)                                                  # forward_model per frame, summed; gradients held by the trainer
        values = {k: float(v) for k, v in loss_dict.items()}                     # the reference's `.item()` per term (:615-616)
        total = sum(values.values())
        if not math.isfinite(total):
            raise FloatingPointError(f"iteration {iteration}: non-finite losses {values}")           # :612
        factor = lr_factor_at(cfg, iteration, max_iter)                          # the scheduler's state when optimizer.step() runs
        trainer.optimizer_step(lr_factor=factor)
        row = {"iteration": iteration + 1, "total_loss": total, **values, "lr": base_lr * factor, "time": time.perf_counter() - t0,
               "data_time": data_time}
        rows.append(row)
        done = iteration + 1
        if log is not None and (done % log_period == 0 or done == max_iter):
            log(row)
        if output_dir and base_state_dict is not None:
            if period > 0 and done % period == 0 and done != max_iter:
                checkpoint.save_checkpoint(os.path.join(output_dir, f"model_{iteration:07d}.pth"), trainer.state_dict(base_state_dict), done)
            if done == max_iter:
                checkpoint.save_checkpoint(os.path.join(output_dir, "model_final.pth"), trainer.state_dict(base_state_dict), done)
        t_data = time.perf_counter()
    model.eval()
    return rows
