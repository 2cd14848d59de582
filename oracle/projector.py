"""ctypes front-end of `oracle/projector.c` (TEST INFRASTRUCTURE ONLY) + intrinsics/pose helpers.

`build()` compiles the C file with gcc into `oracle/_build/liboracle_projector.so`.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_projector.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "projector.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
                               "-o", _SO, src, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def intrinsics_from_vfov(width: int, height: int, vfov: float):
    """`ProjectorUtils.compute_intrinsic_matrix` (Detic/SMNet/projector/core.py:68-77): doubles -> f32."""
    hfov = width / height * vfov
    fx = width / (2.0 * math.tan(hfov / 2.0))
    fy = height / (2.0 * math.tan(vfov / 2.0))
    return (np.float32(fx), np.float32(fy), np.float32(width / 2.0), np.float32(height / 2.0))


def transform3d(xyzhe: np.ndarray) -> np.ndarray:
    """`_transform3D` (Detic/SMNet/projector/core.py:6-34) for one pose, fp32."""
    x, y, z, heading, elev = [np.float32(v) for v in xyzhe]
    cx, sx = np.cos(elev, dtype=np.float32), np.sin(elev, dtype=np.float32)
    cy, sy = np.cos(heading, dtype=np.float32), np.sin(heading, dtype=np.float32)
    T = np.zeros((4, 4), dtype=np.float32)
    T[0] = [cy, sx * sy, cx * sy, x]
    T[1] = [0, cx, -sx, y]
    T[2] = [-sy, cy * sx, cy * cx, z]
    T[3, 3] = 1
    return T


def unproject_world(depth: np.ndarray, T: np.ndarray, fx, fy, cx, cy, proj_shift=(0, 0, 0)) -> np.ndarray:
    lib = _load()
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    H, W = depth.shape
    T = np.ascontiguousarray(T, dtype=np.float32)
    ps = np.ascontiguousarray(proj_shift, dtype=np.float32)
    out = np.empty((H, W, 3), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.oracle_unproject_world(depth.ctypes.data_as(fp), H, W, T.ctypes.data_as(fp),
                               ctypes.c_float(fx), ctypes.c_float(fy), ctypes.c_float(cx), ctypes.c_float(cy),
                               ps.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out


def grid_index(xyz: np.ndarray, map_shift, cell: float, map_w: int, map_h: int, order: int = 0) -> np.ndarray:
    lib = _load()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    shp = xyz.shape[:-1]
    P = int(np.prod(shp))
    ms = np.ascontiguousarray(map_shift, dtype=np.float32)
    out = np.empty((P,), dtype=np.int32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.oracle_grid_index(xyz.ctypes.data_as(fp), ctypes.c_long(P), ms.ctypes.data_as(fp),
                          ctypes.c_float(np.float32(cell)), map_w, map_h, order,
                          out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return out.reshape(shp)


def depth_to_proj_indices(depth, T, intr, proj_shift, map_shift, cell, map_w, map_h, order=0):
    fx, fy, cx, cy = intr
    xyz = unproject_world(depth, T, fx, fy, cx, cy, proj_shift)
    return grid_index(xyz, map_shift, cell, map_w, map_h, order)
