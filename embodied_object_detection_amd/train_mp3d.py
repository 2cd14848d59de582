#!/usr/bin/env python
"""`train_mp3d.py --eval-only` for the MI355X build: same flags and trailing `KEY VALUE` overrides as
`Detic/train_mp3d.py:757-857` (`--config-file --eval-only --num-gpus --num-machines --machine-rank --dist-url --resume`).

    python -m embodied_object_detection_amd.train_mp3d --num-gpus 1 --eval-only \
        --config-file <pkg>/configs/Detic_LCOCOI21k_CLIP_R5021k_640b32_4x_ft4x_max-size_mp3d_recurrent.yaml \
        MODEL.MAP_FEAT_FUSION sum MODEL.MEMORY_TYPE implicit_memory MODEL.MAP_FEATURE_WEIGHT 5

One process per GPU (detectron2 `launch` semantics): either started by `torch.distributed.run` (RANK / WORLD_SIZE in the
environment) or spawned here for `--num-gpus N`; backend `nccl` (= RCCL over xGMI) on GPUs.  Without `--eval-only` the reference's
`do_train` runs first (one process, as `train_mp3d.py:552-553` fixes it for the MP3D loader): `engine/train_loop.py` around the
device step of `modeling/training.py`, then `do_test` on the trained model.  The driver reads the reference's on-disk episodes when `MODEL.TEST_DATA_PATH` holds them
(`memory_data/*.h5`, `sensor_data/*.h5`, `JPEGImages/`: `data/mp3d.py`) and otherwise evaluates the
deterministic synthetic scenes of SURVEY §8d.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch


def default_argument_parser():
    p = argparse.ArgumentParser(description="embodied detector eval (MI355X build)")
    p.add_argument("--config-file", default="", metavar="FILE")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--eval-only", action="store_true")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    p.add_argument("--dist-url", default="tcp://127.0.0.1:29512")
    p.add_argument("--scenes-in-lockstep", type=int, default=1,
                   help="B > 1: every rank runs its scenes B at a time in lock-step; scenes are independent, the AP is the same")
    p.add_argument("--lockstep-schedule", choices=["launches", "streams"], default="launches",
                   help="launches: N = B through every stage, one launch per stage (modeling.lockstep.LockstepScenes); streams: B scene "
                        "objects on their own streams, only the trunk batched (modeling.batched.BatchedSequences)")
    p.add_argument("--synthetic-scenes", type=int, default=2, help="number of synthetic scenes (no real data offline)")
    p.add_argument("--synthetic-frames", type=int, default=40)
    p.add_argument("--synthetic-size", type=int, nargs=2, default=[640, 640])
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VALUE config overrides")
    return p


def do_test(cfg, model, args, rank: int, world: int):
    from .data.synthetic import SyntheticSequence
    from .engine.eval_loop import (episode_offsets, evaluate_gathered, gather_records, inference_on_scenes, rows_needed,
                                   shard_scenes)
    import os
    data_root = str(cfg.MODEL.TEST_DATA_PATH)
    if os.path.isdir(os.path.join(data_root, "memory_data")) and os.path.isdir(os.path.join(data_root, "sensor_data")):
        # the reference's on-disk episodes (train_mp3d.py:393-413): memory_data/*.h5 + sensor_data/*.h5 + JPEGImages/
        from .data.mp3d import Mp3dScenes, SMNetDetectionLoader
        clip_path = cfg.MODEL.ROI_BOX_HEAD.ZEROSHOT_WEIGHT_PATH if cfg.MODEL.MEMORY_TYPE in ("semantic_gt", "map_gt") else None
        loader = SMNetDetectionLoader(data_path=data_root, test_type=cfg.MODEL.TEST_TYPE, clip_path=clip_path,
                                      memory_type=cfg.MODEL.MEMORY_TYPE, semmap_path="")
        ds = Mp3dScenes(loader)
        scenes, offs = ds.shard(rank, world), ds.episode_offsets()
        # same capacity on every rank (the buffer shape is part of the collective): the busiest rank's frame count
        per_rank = [[20] * sum(len(sc.indices) for sc in ds.shard(r, world)) for r in range(world)]     # <= 20 frames per episode
        print(f"[rank {rank}] {len(loader)} episode files in {len(ds)} scenes under {data_root}; this rank: {len(scenes)} scenes")
    else:
        H, W = args.synthetic_size
        mine = shard_scenes(args.synthetic_scenes, rank, world)
        scenes = [SyntheticSequence(s, H=H, W=W, n_frames=args.synthetic_frames) for s in mine]
        offs = dict(enumerate(episode_offsets([args.synthetic_frames] * args.synthetic_scenes)))
        eps = [min(20, args.synthetic_frames - e0) for e0 in range(0, args.synthetic_frames, 20)]           # episodes of 20
        per_rank = [eps * len(shard_scenes(args.synthetic_scenes, r, world)) for r in range(world)]
    # detectron2's `inference_context` (train_mp3d.py:142,366-378): evaluate in eval mode, hand the model back as it came
    was_training = bool(getattr(model, "training", False))
    if was_training:
        model.eval()
    try:
        res = inference_on_scenes(model, scenes, rank, max_rows=max(rows_needed(e) for e in per_rank), scene_episode_offset=offs)
    finally:
        if was_training:
            model.train()
    buf = gather_records(res["records"], rank, world, model.device)
    out = None
    if rank == 0:
        out = evaluate_gathered(buf, int(cfg.MODEL.ROI_HEADS.NUM_CLASSES))
        for name, r in out.items():
            print(f"[eval] {name}: AP {r['AP']:.3f} AP50 {r['AP50']:.3f} AP75 {r['AP75']:.3f} ({r['num_images']} images)")
    print(f"[rank {rank}] {res['frames']} frames in {res['seconds']:.2f} s = {res['frames'] / max(res['seconds'], 1e-9):.1f} frames/s")
    return out


def do_train(cfg, args, rank: int, world: int):
    """`do_train` (train_mp3d.py:509-659) with `DATALOADER.SAMPLER_TRAIN: MP3DLoader`: one process (the reference fixes world_size = 1
    there, :552-553), episodes of `MODEL.TRAIN_DATA_PATH` (memory snapshots under `MODEL.SEMMAP_PATH`), or synthetic episodes when the
    data is not on disk.  Returns the trained model (the same object serves `do_test`)."""
    from . import build_model
    from .checkpoint import fill_missing, load_checkpoint, synthetic_state_dict
    from .engine import train_loop
    from .modeling.training import Trainer
    if world > 1:
        raise NotImplementedError("the MP3D training loader runs on one process (train_mp3d.py:552-553: world_size = 1)")
    num_classes = int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
    # checkpointer.resume_or_load(MODEL.WEIGHTS, resume) (train_mp3d.py:524-527): with --resume the output directory's last
    # checkpoint (weights, optimizer, scheduler, iteration) if there is one, else MODEL.WEIGHTS as weights only
    from .checkpoint import last_checkpoint, load_training_state
    resume_state = None
    weights = str(cfg.MODEL.WEIGHTS) if cfg.MODEL.WEIGHTS else ""
    if args.resume and last_checkpoint(str(cfg.OUTPUT_DIR)) is not None:
        weights = last_checkpoint(str(cfg.OUTPUT_DIR))
        resume_state = load_training_state(weights)
        print(f"[train] resuming from {weights} (iteration {resume_state['iteration']})")
    if weights and os.path.exists(weights):
        sd = fill_missing(load_checkpoint(weights, num_classes)[0], 0, num_classes)
    else:
        sd = synthetic_state_dict(0, num_classes, cfg.MODEL.ROI_BOX_HEAD.ZEROSHOT_WEIGHT_PATH)
    model = build_model(cfg, sd)
    trainer = Trainer(model, sd)
    data_root = str(cfg.MODEL.TRAIN_DATA_PATH)
    ims = max(int(cfg.SOLVER.IMS_PER_BATCH) // world, 1)
    if os.path.isdir(os.path.join(data_root, "memory_data")) and os.path.isdir(os.path.join(data_root, "sensor_data")):
        from .data.mp3d import SMNetDetectionLoader, collate_smnet, map_mp3d_batch_to_coco
        clip_path = cfg.MODEL.ROI_BOX_HEAD.ZEROSHOT_WEIGHT_PATH if cfg.MODEL.MEMORY_TYPE in ("semantic_gt", "map_gt") else None
        loader = SMNetDetectionLoader(data_path=data_root, clip_path=clip_path, memory_type=cfg.MODEL.MEMORY_TYPE,
                                      semmap_path=str(cfg.MODEL.SEMMAP_PATH))
        # the reference's DataLoader: two forked worker processes read and decode the episodes ahead (train_mp3d.py:563-572)
        batches = train_loop.training_batches(loader, ims, seed=0, collate=collate_smnet, workers=int(cfg.DATALOADER.NUM_WORKERS_TRAIN_MP3D))
        map_batch = map_mp3d_batch_to_coco
    else:
        from .data.synthetic import SyntheticTrainingEpisodes
        H, W = args.synthetic_size
        ds = SyntheticTrainingEpisodes(args.synthetic_scenes, H=H, W=W, n_frames=min(args.synthetic_frames, 20))
        batches, map_batch = train_loop.training_batches(ds, ims, seed=0), None
    os.makedirs(str(cfg.OUTPUT_DIR), exist_ok=True)
    rows = train_loop.do_train(cfg, model, trainer, batches, resume_state=resume_state, output_dir=str(cfg.OUTPUT_DIR), base_state_dict=sd,
                               map_batch=map_batch, do_test=lambda: do_test(cfg, model, args, rank, world),
                               log=lambda r: print("[train] " + json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()})))
    if rows:
        print(f"[train] {len(rows)} iterations, total loss {rows[0]['total_loss']:.4f} -> {rows[-1]['total_loss']:.4f}; "
              f"{sum(r['time'] for r in rows) / len(rows) * 1e3:.1f} ms per iteration; train-mode proposal lists {trainer.fm.pre} / {trainer.fm.post}")
    return model


def main(args, rank: int = 0, world: int = 1, local_rank: int = 0):
    from . import build_model, setup_cfg
    opts = list(args.opts or [])
    if opts and opts[0] == "--":
        opts = opts[1:]
    opts += ["MODEL.DEVICE", f"cuda:{local_rank}"]
    cfg = setup_cfg(args.config_file or None, opts)
    if not args.eval_only:
        model = do_train(cfg, args, rank, world)                   # train_mp3d.py:740-741: do_train, then do_test on the trained model
        return do_test(cfg, model, args, rank, world)
    if args.scenes_in_lockstep > 1 and args.lockstep_schedule == "launches":
        from .modeling.lockstep import LockstepScenes
        model = LockstepScenes(cfg, args.scenes_in_lockstep)
    elif args.scenes_in_lockstep > 1:
        from .modeling.batched import BatchedSequences
        model = BatchedSequences(cfg, args.scenes_in_lockstep)
    else:
        model = build_model(cfg)
    return do_test(cfg, model, args, rank, world)


def _worker(local_rank: int, args, world: int):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", args.dist_url.rsplit(":", 1)[-1])
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=local_rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    try:
        main(args, local_rank, world, local_rank)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def launch(args):
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        import torch.distributed as dist
        rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{lr}"))
        try:
            return main(args, rank, world, lr)
        finally:
            dist.barrier()
            dist.destroy_process_group()
    if args.num_gpus > 1:
        import torch.multiprocessing as mp
        mp.spawn(_worker, args=(args, args.num_gpus), nprocs=args.num_gpus, join=True)
        return None
    return main(args)


if __name__ == "__main__":
    launch(default_argument_parser().parse_args())
