"""`do_train` of the reference's driver, host side (`Detic/train_mp3d.py:509-659`) around the device step of `modeling/training.py`.

The reference's iteration -- `data = map_mp3d_batch_to_coco(data); loss_dict = model(data); losses.backward(); optimizer.step();
scheduler.step(); periodic_checkpointer.step(iteration)` -- with the same sampler semantics (detectron2's `TrainingSampler`: an
infinite stream of seeded shuffles of the episode indices, `IMS_PER_BATCH` episodes per iteration, `drop_last`), the same schedule
(`WarmupCosineLR` as `solver.warmup_cosine_lr_factor` restates it), the same finite-loss assertion (:612), the same checkpoint
rhythm and the same resume rule.  **Pinned by the reference's own loop**: `tests/golden/gen_golden_io.py::gen_train_driver` runs
`train_mp3d.py:509-659` with a stub model / optimizer / checkpointer on the written dataset; `tests/test_io_golden.py` holds this
loop to the recorded episodes, iteration numbers, learning rates, checkpoint names and stored iterations, `do_test` calls and writer
rhythm (`tests/golden/mp3d_train_driver.json`).  Three things the reference's loop does that one would not guess (all reproduced):
  * `iteration` is incremented before the loop body uses it (:605), so `PeriodicCheckpointer.step(iteration)` sees a 1-based number:
    `model_{i:07d}.pth` appears when (i + 1) % CHECKPOINT_PERIOD == 0 with `iteration` = i = the number of finished iterations, and
    `model_final.pth` is written twice, after max_iter - 1 and after max_iter iterations;
  * a resumed run starts at stored iteration + 1 (:524-525) and the body adds one again: the iteration numbered stored + 1 never
    runs -- a run resumed from `model_final` written at max_iter - 1 does nothing;
  * the schedule is built from SOLVER.MAX_ITER (`build_lr_scheduler`, :519) even when SOLVER.TRAIN_ITER caps the loop (:529).
What is not here: the AMP GradScaler (`Trainer` refuses FP16: True), TensorBoard / JSON writers (a `log` callable takes their rows at
the writers' rhythm, :647-650).  The loader's worker processes (:563-572) are `training_batches(..., workers=2)`.
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


class _TrainingSampler(torch.utils.data.Sampler):
    """`training_sampler` as a torch Sampler object (what `DataLoader(sampler=TrainingSampler(len(t_loader)))` is handed, :552)."""

    def __init__(self, size: int, seed: int = 0, shuffle: bool = True):
        self.size, self.seed, self.shuffle = size, seed, shuffle

    def __iter__(self):
        return training_sampler(self.size, self.seed, self.shuffle)


def training_batches(dataset, ims_per_batch: int, seed: int = 0, shuffle: bool = True, collate: Optional[Callable] = None,
                     workers: int = 0) -> Iterator[List]:
    """The DataLoader of `train_mp3d.py:563-572`: `ims_per_batch` episodes per iteration drawn by `training_sampler`, collated by
    `collate_smnet` (a list of episodes; an episode is the loader's list of frame records).  `workers` > 0: the reference's own
    arrangement -- a torch DataLoader with that many forked worker PROCESSES reading and decoding episodes ahead (`num_workers=2`,
    `multiprocessing_context='fork'`, `drop_last`, :563-572); the batches and their order are those of `workers = 0` (the workers
    touch files and host memory only, never the GPU)."""
    if ims_per_batch < 1 or len(dataset) < 1:
        raise ValueError("training needs at least one episode per batch and a non-empty dataset")
    if workers > 0:
        dl = torch.utils.data.DataLoader(dataset, batch_size=ims_per_batch, num_workers=int(workers), drop_last=True,
                                         sampler=_TrainingSampler(len(dataset), seed, shuffle), multiprocessing_context="fork",
                                         collate_fn=collate if collate is not None else (lambda b: list(b)), pin_memory=False)
        yield from dl
        return
    it = training_sampler(len(dataset), seed, shuffle)
    while True:
        batch = [dataset[next(it)] for _ in range(ims_per_batch)]
        yield collate(batch) if collate is not None else batch


def lr_factor_at(cfg, step: int) -> float:
    """The schedule's factor when `step` scheduler steps have been taken (`build_lr_scheduler(cfg, optimizer)`, train_mp3d.py:519: built
    from SOLVER.MAX_ITER, whatever SOLVER.TRAIN_ITER says)."""
    from .. import solver
    s = cfg.SOLVER
    if str(s.LR_SCHEDULER_NAME) != "WarmupCosineLR":
        raise NotImplementedError("SOLVER.LR_SCHEDULER_NAME: WarmupCosineLR (Base-C2_L_R5021k_640b64_4x_recurrent.yaml:64)")
    return solver.warmup_cosine_lr_factor(step, int(s.MAX_ITER), int(s.WARMUP_ITERS), float(s.WARMUP_FACTOR), str(s.WARMUP_METHOD))


def do_train(cfg, model, trainer, batches: Iterator[List], *, resume_state: Optional[Dict] = None, output_dir: Optional[str] = None,
             base_state_dict: Optional[Dict[str, torch.Tensor]] = None, map_batch: Optional[Callable] = None,
             log: Optional[Callable[[Dict], None]] = None, do_test: Optional[Callable[[], None]] = None,
             on_save: Optional[Callable[[str, int], None]] = None) -> List[Dict]:
    """-> one row per iteration {iteration, total_loss, <loss names>, lr, time, data_time}.  `batches`: iterator of loader batches
    (`training_batches`); `map_batch`: `map_mp3d_batch_to_coco` (None: the batches already hold frame dicts).

    `resume_state` (`--resume`: what `checkpoint.load_training_state` reads from the output directory's last checkpoint; None = a
    fresh run, also when MODEL.WEIGHTS holds an 'iteration', :526-527): {'iteration', 'optimizer', 'scheduler'} -- the optimizer's
    moments and step counts and the scheduler's position are restored, the loop starts at iteration + 1.

    Checkpoints (when `output_dir` and `base_state_dict` are given) in DetectionCheckpointer's format with the optimizer's and the
    scheduler's state (`checkpoint.save_checkpoint`); `on_save(name, iteration)` is told about every save; `do_test()` runs every
    TEST.EVAL_PERIOD iterations (:636-640); `log(row)` at the writers' rhythm (:647-650)."""
    from .. import checkpoint
    s = cfg.SOLVER
    max_iter = int(s.MAX_ITER) if int(s.get("TRAIN_ITER", -1)) < 0 else int(s.TRAIN_ITER)              # train_mp3d.py:529
    period = int(s.CHECKPOINT_PERIOD)
    eval_period = int(cfg.TEST.EVAL_PERIOD)
    base_lr = float(s.BASE_LR)
    start_iter, sched_step = 0, 0
    if resume_state is not None:
        start_iter = int(resume_state["iteration"]) + 1                                                 # :524-525
        sched_step = int(resume_state["scheduler"]["last_epoch"])
        if resume_state.get("optimizer") is not None and hasattr(trainer, "load_optimizer_state"):
            trainer.load_optimizer_state(resume_state["optimizer"])
    model.train()
    rows: List[Dict] = []

    def save(name: str, iteration: int):
        if on_save is not None:
            on_save(name, iteration)
        if output_dir and base_state_dict is not None:
            opt_state = trainer.optimizer_state() if hasattr(trainer, "optimizer_state") else None
            checkpoint.save_checkpoint(os.path.join(output_dir, name), trainer.state_dict(base_state_dict), iteration,
                                       optimizer=opt_state, scheduler={"last_epoch": sched_step})

    t_data = time.perf_counter()
    for data, iteration in zip(batches, range(start_iter, max_iter)):
        if map_batch is not None:
            data = map_batch(data)
        data_time = time.perf_counter() - t_data
        t0 = time.perf_counter()
        iteration = iteration + 1                                                # :605 -- everything below sees the 1-based number
        loss_dict = model(data)                                                  # forward_model per frame, summed; gradients held by the trainer
        values = {k: float(v) for k, v in loss_dict.items()}                     # the reference's `.item()` per term (:615-616)
        total = sum(values.values())
        if not math.isfinite(total):
            raise FloatingPointError(f"iteration {iteration}: non-finite losses {values}")           # :612
        factor = lr_factor_at(cfg, sched_step)                                   # the scheduler's state when optimizer.step() runs
        trainer.optimizer_step(lr_factor=factor)
        row = {"iteration": iteration, "total_loss": total, **values, "lr": base_lr * factor, "time": time.perf_counter() - t0,
               "data_time": data_time}
        rows.append(row)
        sched_step += 1                                                          # scheduler.step() (:633)
        if do_test is not None and eval_period > 0 and iteration % eval_period == 0 and iteration != max_iter:    # :636-640
            do_test()
            model.train()                                                        # detectron2's inference_context restores the mode
        if log is not None and iteration - start_iter > 5 and (iteration % 20 == 0 or iteration == max_iter):      # :647-650
            log(row)
        # PeriodicCheckpointer.step(iteration) (:654; fvcore): the periodic file, then the final one
        if period > 0 and (iteration + 1) % period == 0:
            save(f"model_{iteration:07d}.pth", iteration)
        if iteration >= max_iter - 1:
            save("model_final.pth", iteration)
        t_data = time.perf_counter()
    return rows
