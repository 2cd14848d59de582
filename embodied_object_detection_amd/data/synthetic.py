"""Deterministic synthetic sequences with the frame-dict schema of the reference loader.

Schema: `SMNetDetectionLoader.__getitem__` + `map_mp3d_batch_to_coco` (`Detic/SMNet/loader.py:171-308`,
`Detic/train_mp3d.py:452-507`): one item = one episode of <= 20 frame dicts with keys image (u8 [3,H,W] RGB),
height, width, proj_indices (np.int32 [H,W,1]), memory (only shape[0] is used), memory_reset, sequence_name,
observations, file_name, image_id, instances (GT, eval only).  Generator recipe: SURVEY.md §8d.

`proj_indices` come from the build's own depth un-projection + grid indexing (a1+a2).  The projector is injected:
the default is the HIP kernel; CPU-only host-logic tests inject the oracle's C restatement.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

VFOV = 67.5 * math.pi / 180.0
EPISODE_LEN = 20


def intrinsics_from_vfov(width: int, height: int, vfov: float = VFOV):
    """fx, fy, cx, cy as `ProjectorUtils.compute_intrinsic_matrix` (Detic/SMNet/projector/core.py:68-77)."""
    hfov = width / height * vfov
    fx = width / (2.0 * math.tan(hfov / 2.0))
    fy = height / (2.0 * math.tan(vfov / 2.0))
    return (np.float32(fx), np.float32(fy), np.float32(width / 2.0), np.float32(height / 2.0))


def transform3d(xyzhe) -> np.ndarray:
    """Camera-to-world matrix from (x, y, z, heading, elevation) (`_transform3D`, core.py:6-34), fp32."""
    x, y, z, heading, elev = [np.float32(v) for v in xyzhe]
    cx, sx = np.cos(elev, dtype=np.float32), np.sin(elev, dtype=np.float32)
    cy, sy = np.cos(heading, dtype=np.float32), np.sin(heading, dtype=np.float32)
    T = np.zeros((4, 4), dtype=np.float32)
    T[0] = [cy, sx * sy, cx * sy, x]
    T[1] = [0, cx, -sx, y]
    T[2] = [-sy, cy * sx, cy * cx, z]
    T[3, 3] = 1
    return T


def hip_projector(device="cuda:0") -> Callable:
    from .. import ops

    def fn(depth: np.ndarray, T, intr, proj_shift, map_shift, cell, map_w, map_h, order=0) -> np.ndarray:
        d = torch.from_numpy(np.ascontiguousarray(depth, dtype=np.float32)).to(device)
        idx = ops.unproject_grid_index(d, T, intr, proj_shift, map_shift, cell, map_w, map_h, order)
        return idx.cpu().numpy()
    return fn


class SyntheticSequence:
    """One scene: `n_frames` frames in episodes of 20; `memory_reset` only on the very first frame
    (TEST_TYPE 'default', loader.py:289-293)."""

    def __init__(self, seq_id: int, H: int = 640, W: int = 640, n_frames: int = 100, map_w: int = 200, map_h: int = 200,
                 cell: float = 0.2, map_shift=(-5.0, 0.0, -5.0), projector: Optional[Callable] = None, num_classes: int = 20,
                 max_gt: int = 6):
        self.seq_id, self.H, self.W, self.n_frames = seq_id, H, W, n_frames
        self.map_w, self.map_h, self.cell = map_w, map_h, float(cell)
        self.map_shift = np.asarray(map_shift, dtype=np.float32)
        self.n_cells = map_w * map_h
        self.name = f"synthetic_{seq_id:05d}"
        self.projector = projector
        self.num_classes, self.max_gt = num_classes, max_gt
        self._intr = intrinsics_from_vfov(W, H)
        g = torch.Generator().manual_seed(1234 + seq_id)
        # camera random walk inside a 30 m x 30 m area, height 1.5 m
        pos = torch.rand((2,), generator=g) * 20.0 + 5.0
        heading = float(torch.rand((1,), generator=g)) * 2 * math.pi
        self.poses = []
        for _ in range(n_frames):
            heading += float(torch.rand((1,), generator=g) * 0.4 - 0.2)
            step = 0.25
            pos = torch.clamp(pos + step * torch.tensor([math.sin(heading), math.cos(heading)]), 1.0, 29.0)
            self.poses.append((float(pos[0]), 1.5, float(pos[1]), heading, math.pi))
        self._g_seed = 99991 * (seq_id + 1)

    def frame(self, i: int) -> Dict:
        g = torch.Generator().manual_seed(self._g_seed + i)
        H, W = self.H, self.W
        image = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g)
        low = torch.rand((1, 1, max(H // 32, 2), max(W // 32, 2)), generator=g) * 2 - 1
        smooth = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)[0, 0]
        depth = torch.clamp(3.0 + 2.0 * smooth, 0.5, 10.0).numpy().astype(np.float32)
        T = transform3d(self.poses[i])
        if self.projector is None:
            self.projector = hip_projector()
        proj = self.projector(depth, T, self._intr, (0.0, 0.0, 0.0), self.map_shift, self.cell, self.map_w, self.map_h, 0)
        proj = np.ascontiguousarray(proj, dtype=np.int32).reshape(H, W, 1)
        ngt = int(torch.randint(1, self.max_gt + 1, (1,), generator=g))
        ctr = torch.rand((ngt, 2), generator=g) * torch.tensor([float(W), float(H)])
        size = torch.rand((ngt, 2), generator=g) * torch.tensor([W * 0.4, H * 0.4]) + 16
        gt = torch.cat([ctr - size / 2, ctr + size / 2], dim=1)
        gt[:, 0::2] = gt[:, 0::2].clamp(0, W)
        gt[:, 1::2] = gt[:, 1::2].clamp(0, H)
        gt_classes = torch.randint(0, self.num_classes, (ngt,), generator=g)
        return {
            "image": image, "height": H, "width": W, "proj_indices": proj,
            "memory": np.zeros((self.n_cells, 1), dtype=np.float32),   # only shape[0] is used (custom_rcnn.py:475-477)
            "memory_reset": i == 0, "sequence_name": self.name, "observations": None,
            "file_name": f"{self.name}/{i:04d}.jpg", "image_id": self.seq_id * 100000 + i,
            "depth": depth, "pose": self.poses[i],
            "instances": {"gt_boxes": gt, "gt_classes": gt_classes},
        }

    def episodes(self):
        for e0 in range(0, self.n_frames, EPISODE_LEN):
            yield [self.frame(i) for i in range(e0, min(self.n_frames, e0 + EPISODE_LEN))]


class SyntheticTrainingEpisodes:
    """Episodes for the training loop when no data is on disk: the frames of `SyntheticSequence` with what the training loader adds
    (loader.py:199-223 with MODEL.SEMMAP_PATH set): an accumulated memory `memory` [n_cells,512] with its observation counts
    `observations` [n_cells] (seeded noise scaled by the counts).  `dataset[i]` = episode i as a list of frame dicts."""

    def __init__(self, n_scenes: int, H: int = 640, W: int = 640, n_frames: int = 20, **kw):
        self.scenes = [SyntheticSequence(s, H=H, W=W, n_frames=n_frames, **kw) for s in range(n_scenes)]
        self.n_frames = n_frames

    def __len__(self) -> int:
        return len(self.scenes)

    def __getitem__(self, index: int):
        sc = self.scenes[index]
        g = torch.Generator().manual_seed(7919 * (index + 1))
        obs = torch.randint(0, 6, (sc.n_cells,), generator=g).float()
        mem = (torch.randn((sc.n_cells, 512), generator=g) * obs.clamp(min=1.0)[:, None]).numpy()
        frames = []
        for i in range(self.n_frames):
            f = sc.frame(i)
            f["memory"], f["observations"] = mem, obs.numpy()
            frames.append(f)
        return frames


def build_synthetic_dataset(n_sequences: int, **kw) -> List[SyntheticSequence]:
    return [SyntheticSequence(s, **kw) for s in range(n_sequences)]
