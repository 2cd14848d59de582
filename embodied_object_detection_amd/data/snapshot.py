"""Memory snapshot dump / load.

Writer: `CustomRCNNRecurrent.forward` under `MODEL.TEST_SAVE_SEMMAP` (custom_rcnn.py:518-530) stores, after the first frame of
every inner sequence, three datasets named `semmap` (int32 [N]), `impicit_memory` [sic] (float32 [N,512]) and `observations`
(float32 [N]) in `<OUTPUT_DIR>/memory/<sequence_name>`.  Reader: `SMNetDetectionLoader` (SMNet/loader.py:216-223) loads the same
three names from `MODEL.SEMMAP_PATH/<file>` and shifts the labels by +1 (empty space -1 -> 0).

The container is HDF5 like the reference's (written / read through `data/h5io.py`, libhdf5 via ctypes; the reference uses h5py) at
exactly the reference's path.  Only when no libhdf5 can be loaded the same datasets go into a NumPy `.npz` archive next to that
path (`<name>.npz`); the reader accepts both.
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
    from . import h5io
    d = os.path.join(output_dir, "memory")
    os.makedirs(d, exist_ok=True)
    if h5io.available():
        path = os.path.join(d, sequence_name)
        with h5io.H5File(path, "w") as f:
            f.write(KEY_SEMMAP, semmap, dtype=np.int32)
            f.write(KEY_MEMORY, implicit_memory, dtype=np.float32)
            f.write(KEY_OBS, observations, dtype=np.float32)
        return path
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
    exact = os.path.join(semmap_path, file)
    if os.path.exists(exact) and not exact.endswith(".npz"):
        from .h5io import H5File
        with H5File(exact) as f:
            return {"semmap_real": f.read(KEY_SEMMAP) + 1, "implicit_memory": f.read(KEY_MEMORY), "observations": f.read(KEY_OBS)}
    with np.load(snapshot_path(semmap_path, file)) as z:
        return {"semmap_real": np.array(z[KEY_SEMMAP]) + 1, "implicit_memory": np.array(z[KEY_MEMORY]),
                "observations": np.array(z[KEY_OBS])}
