"""Robot online front-end: second caller of the model boundary.

Mirrors the per-frame preparation of `Detic/robot_demo.py:489-534` and `EmbodiedPredictor.__call__`
(`Detic/detic/predictor.py:406-439`): nearest-timestamp association of RGB / depth / pose files, depth in millimetres -> metres,
the fixed RealSense intrinsics, the axis-swapped camera transform `T @ R` (`robot_demo.py:40-90`), grid-cell indexing with the
robot's `x * map_h + y` ordering (`robot_demo.py:533`), and the frame dict the model takes.  `RobotFrontEnd` takes arrays; `RobotRun` reads a recorded run from disk (16-bit depth PNGs and
RGB images through Pillow, poses through numpy).  The un-projection + indexing runs in the HIP kernel (`order=1`).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# `ProjectorUtils.compute_intrinsic_matrix` of robot_demo.py:124-126 (fixed K, not derived from the fov)
ROBOT_INTRINSICS = (np.float32(380.3127746582031), np.float32(379.828857421875), np.float32(315.81829833984375),
                    np.float32(250.9555206298828))
ROBOT_MAP_SHIFT = (-13.0, 0.0, -13.0)     # robot_demo.py:476
ROBOT_RES = 0.2                            # robot_demo.py:470
ROBOT_MAP_W = ROBOT_MAP_H = math.ceil(40 / ROBOT_RES)   # 200 x 200 (robot_demo.py:471-474)
ROBOT_CAMERA_HEIGHT = 0.65
ROBOT_ELEVATION = math.pi + 0.06


def nearest_by_timestamp(stamp: int, candidates: Sequence[str]) -> str:
    """`min(files, key=|int(stem) - t|)` (robot_demo.py:493-496); first minimum wins, like Python's `min`."""
    return min(candidates, key=lambda x: abs(int(x.split(".")[0]) - int(stamp)))


def robot_transform(pose_xyt: Sequence[float]) -> np.ndarray:
    """`_transform3D(xyzhe) @ R` of robot_demo.py:40-90 with xyzhe = (x, 0.65, y, -theta, pi + 0.06) (robot_demo.py:519), fp32."""
    x, y, th = [np.float32(v) for v in pose_xyt[:3]]
    heading, elev = np.float32(-th), np.float32(ROBOT_ELEVATION)
    cx, sx = np.cos(elev, dtype=np.float32), np.sin(elev, dtype=np.float32)
    cy, sy = np.cos(heading, dtype=np.float32), np.sin(heading, dtype=np.float32)
    T = np.zeros((4, 4), dtype=np.float32)
    T[0] = [cy, sx * sy, cx * sy, x]
    T[1] = [0, cx, -sx, np.float32(ROBOT_CAMERA_HEIGHT)]
    T[2] = [-sy, cy * sx, cy * cx, y]
    T[3, 3] = 1
    R = np.zeros((4, 4), dtype=np.float32)
    R[0, 2] = R[1, 1] = R[2, 0] = R[3, 3] = 1
    return (T @ R).astype(np.float32)


class RobotFrontEnd:
    """Builds model inputs from raw robot frames; one instance per run (the memory grid is fixed at 200 x 200 @ 0.2 m)."""

    def __init__(self, projector: Optional[Callable] = None, map_w: int = ROBOT_MAP_W, map_h: int = ROBOT_MAP_H,
                 res: float = ROBOT_RES, map_shift=ROBOT_MAP_SHIFT, sequence_name: str = "robot"):
        self.projector = projector
        self.map_w, self.map_h, self.res = map_w, map_h, float(res)
        self.map_shift = np.asarray(map_shift, dtype=np.float32)
        self.n_cells = map_w * map_h
        self.sequence_name = sequence_name
        self._first = True

    def frame(self, rgb_hwc_u8: np.ndarray, depth_mm: np.ndarray, pose_xyt: Sequence[float], file_name: str = "") -> Dict:
        if self.projector is None:
            from .synthetic import hip_projector
            self.projector = hip_projector()
        H, W = depth_mm.shape
        # robot_demo.py:515-517: uint16 millimetres / 1000 in float64, then FloatTensor (one rounding to fp32)
        depth_m = (np.asarray(depth_mm, dtype=np.float64) / 1000).astype(np.float32)
        T = robot_transform(pose_xyt)
        proj = self.projector(depth_m, T, ROBOT_INTRINSICS, (0.0, 0.0, 0.0), self.map_shift, self.res, self.map_w, self.map_h, 1)
        image = torch.from_numpy(np.ascontiguousarray(np.asarray(rgb_hwc_u8).transpose(2, 0, 1)))
        reset, self._first = self._first, False
        return {"image": image, "height": H, "width": W, "proj_indices": np.ascontiguousarray(proj, dtype=np.int32).reshape(H, W, 1),
                "memory": np.zeros((self.n_cells, 1), dtype=np.float32), "memory_reset": reset,
                "sequence_name": self.sequence_name, "observations": None, "file_name": file_name}

    def robot_cell(self, pose_xyt: Sequence[float]) -> Tuple[int, int]:
        """Robot position on the map (robot_demo.py:536-538)."""
        p = (np.asarray(pose_xyt[:2], dtype=np.float32) - self.map_shift[[0, 2]]) / np.float32(self.res)
        q = np.rint(p).astype(np.int64)
        return int(q[0]), int(q[1])


def read_depth_mm(path: str) -> np.ndarray:
    """`cv2.imread(path, cv2.IMREAD_ANYDEPTH)` (robot_demo.py:498): the 16-bit depth PNG as uint16 millimetres, via Pillow."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im)
    if a.ndim != 2:
        raise ValueError(f"{path}: expected a single-channel depth image, got shape {a.shape}")
    return a.astype(np.uint16) if a.dtype != np.uint16 else a


def read_rgb(path: str) -> np.ndarray:
    """The Detic loader's own decode, inlined at robot_demo.py:503-507: Pillow, EXIF orientation, RGB uint8 HWC."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        return np.asarray(ImageOps.exif_transpose(im).convert("RGB"))


class RobotRun:
    """One recorded robot run on disk (`<root>/images`, `<root>/depth`, `<root>/pose`, file stems = timestamps): every second RGB
    image (~10 Hz) with the depth image and the odometry sample closest in time (robot_demo.py:485-499), turned into model frames
    by a `RobotFrontEnd`."""

    def __init__(self, root: str, front_end: Optional[RobotFrontEnd] = None, every: int = 2):
        import os
        self.root = root
        self.images = sorted(os.listdir(os.path.join(root, "images")))
        self.depth = sorted(os.listdir(os.path.join(root, "depth")))
        self.pose = sorted(os.listdir(os.path.join(root, "pose")))
        self.every = every
        self.front_end = front_end or RobotFrontEnd(sequence_name=os.path.basename(os.path.normpath(root)))

    def __len__(self) -> int:
        return len(self.images[::self.every])

    def __iter__(self):
        import os
        for image in self.images[::self.every]:
            stamp = image.split(".")[0]
            d = nearest_by_timestamp(stamp, self.depth)
            p = nearest_by_timestamp(stamp, self.pose)
            depth_mm = read_depth_mm(os.path.join(self.root, "depth", d))
            pose = np.load(os.path.join(self.root, "pose", p))
            rgb = read_rgb(os.path.join(self.root, "images", image))
            f = self.front_end.frame(rgb, depth_mm, pose, file_name=image)
            f["depth_file"], f["pose_file"] = d, p
            yield f
