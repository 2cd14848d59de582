"""Memory snapshot dump / load.

Writer: `CustomRCNNRecurrent.forward` under `MODEL.TEST_SAVE_SEMMAP` (custom_rcnn.py:518-530) stores, after the first frame of
every inner sequence, three datasets named `semmap` (int32 [N]), `impicit_memory` [sic] (float32 [N,512]) and `observations`
(float32 [N]) in `<OUTPUT_DIR>/memory/<sequence_name>`.  Reader: `SMNetDetectionLoader` (SMNet/loader.py:216-223) loads the same
three names from `MODEL.SEMMAP_PATH/<file>` and shifts the labels by +1 (empty space -1 -> 0).

The reference container is HDF5 through h5py, which this image does not have; the datasets keep their names, dtypes and shapes
inside a NumPy `.npz` archive instead (`np.load(path)[name]` replaces `h5py.File(path)[name]`).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np

KEY_SEMMAP = "semmap"
KEY_MEMORY = "impicit_memory"      # the reference's spelling (custom_rcnn.py:529, loader.py:218)
KEY_OBS = "observations"


def snapshot_path(directory: str, sequence_name: str) -> str:
    name = sequence_name if sequence_name.endswith(".npz") else sequence_name + ".npz"
    return os.path.join(directory, name)


def write_snapshot(output_dir: str, sequence_name: str, semmap: np.ndarray, implicit_memory: np.ndarray,
                   observations: np.ndarray) -> str:
    """custom_rcnn.py:520-530; returns the path written."""
    d = os.path.join(output_dir, "memory")
    os.makedirs(d, exist_ok=True)
    path = snapshot_path(d, sequence_name)
    with open(path, "wb") as f:
        np.savez(f, **{KEY_SEMMAP: np.asarray(semmap, dtype=np.int32), KEY_MEMORY: np.asarray(implicit_memory, dtype=np.float32),
                       KEY_OBS: np.asarray(observations, dtype=np.float32)})
    return path


def read_snapshot(semmap_path: str, file: str, fallback_memory: Optional[np.ndarray] = None) -> Dict[str, Optional[np.ndarray]]:
    """loader.py:214-227: `semmap_real` (+1 shifted), `implicit_memory`, `observations`; when `semmap_path` does not exist the
    loader falls back to the offline memory and `None` for the other two."""
    if not os.path.exists(semmap_path):
        return {"semmap_real": None, "implicit_memory": fallback_memory, "observations": None}
    with np.load(snapshot_path(semmap_path, file)) as z:
        return {"semmap_real": np.array(z[KEY_SEMMAP]) + 1, "implicit_memory": np.array(z[KEY_MEMORY]),
                "observations": np.array(z[KEY_OBS])}
