"""On-disk episode reader of the reference: `memory_data/*.h5` + `sensor_data/*.h5` + `JPEGImages/`.

Mirrors `SMNetDetectionLoader` (`Detic/SMNet/loader.py:56-308`), `collate_smnet` (`:51-54`) and `map_mp3d_batch_to_coco`
(`Detic/train_mp3d.py:452-507`): same constructor arguments, same file ordering and `longterm` duplication (`:97-117`), same
datasets (`memory_features`, `semmap_gt`, `proj_indices`; `rgb`, `segmentation_data`, `detection_data`; optional snapshot
`semmap`, `impicit_memory`, `observations`), same `detection_data` parsing, class filter, XYWH -> XYXY, JPEG decode with EXIF
orientation, `memory_reset` rule (`:289-293`) and the same keys in the returned frame dicts.  HDF5 goes through
`data/h5io.py` (libhdf5 via ctypes; the reference uses h5py), JPEG through Pillow like the reference.

`Mp3dScenes` groups the episode files by scene for `engine/eval_loop.inference_on_scenes` (scene -> rank sharding, §8e).
"""
from __future__ import annotations

import ast
import json
import os
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from ..structures import Boxes, Instances
from .h5io import H5File

# object_lvis subset evaluated by the reference (loader.py:133) and its map-GT class mapping (:136)
CLASS_IDS = [0, 2, 3, 4, 5, 6, 7, 9, 12, 13, 14, 15, 16, 17, 19]
SMNET_CLASS_MAPPING = [0, 11, 17, 1, 14, 4, 13, 10, 16, 6, 0, 0, 18]


def collate_smnet(batch):
    return batch


def episode_sort_key(name: str):
    """loader.py:97-105: (everything before the last '_' token, integer of the last token)."""
    parts = name.split("_")
    prefix = "".join(p + "_" for p in parts[:-1])
    return prefix, int(parts[-1].split(".")[0])


def longterm_file_list(files: Sequence[str]) -> List[str]:
    """loader.py:108-117: chunks of 50 episodes, every chunk twice in a row, and the first episode of each repeat replaced by
    the one before it (so that the repeat does not start with a memory reset)."""
    chunks = [list(files[i:i + 50]) for i in range(0, len(files), 50)]
    chunks = sorted(chunks * 2)
    flat = [f for ch in chunks for f in ch]
    for j in range(50, len(flat), 100):
        flat[j] = flat[j - 1]
    return flat


def parse_detection_data(raw: bytes):
    """One `detection_data` record (`str(dict)` written by build_data.py:245) -> (file_name, XYXY boxes, classes), unfiltered
    (loader.py:247-253)."""
    s = raw.decode().replace("'", "\"")
    file_name = s.split('"file_name": ')[1].split(', "image": ')[0]
    gt_box, gt_class = s.split('"gt_boxes": ')[1].split(', "gt_classes": ')
    gt_class = ast.literal_eval(gt_class[:-1])
    gt_box = ast.literal_eval(gt_box)
    gt_box = [[b[0], b[1], b[2] + b[0], b[3] + b[1]] for b in gt_box]
    return file_name[1:-1], gt_box, gt_class


def read_image_rgb(path: str) -> np.ndarray:
    """d2 `read_image(format="RGB")` as inlined at loader.py:272-277: PIL decode, EXIF orientation, RGB uint8 HWC."""
    from PIL import Image, ImageOps
    with open(path, "rb") as f:
        im = Image.open(f)
        im = ImageOps.exif_transpose(im)
        return np.asarray(im.convert("RGB"))


class SMNetDetectionLoader:
    def __init__(self, data_path: str = "sensor_data", test_type: str = "default", clip_path: Optional[str] = None,
                 memory_type: str = "", semmap_path: str = "gt", semmap_gt_info: str = "SMNet/semmap_GT_info.json"):
        self.clip_path = clip_path
        self.memory_type = memory_type
        self.memory_path = os.path.join(data_path, "memory_data")
        self.data_path = os.path.join(data_path, "sensor_data")
        self.image_root = os.path.join(data_path, "JPEGImages")
        self.test_type = test_type
        self.max_sequence_length = 20
        self.semmap_path = semmap_path
        self.semmap_gt_info = json.load(open(semmap_gt_info)) if os.path.exists(semmap_gt_info) else None   # loaded, never read
        self.files = sorted(os.listdir(self.memory_path), key=episode_sort_key)
        if self.test_type == "longterm":
            self.files = longterm_file_list(self.files)
        self.envs = [x.split(".")[0] for x in self.files]
        self.class_ids = list(CLASS_IDS)
        self.smnet_class_mapping = list(SMNET_CLASS_MAPPING)
        if self.clip_path:
            self.clip_embeddings = np.load(self.clip_path)
        assert len(self.files) > 0
        self.available_idx = list(range(len(self.files)))

    def __len__(self) -> int:
        return len(self.available_idx)

    def __getitem__(self, index: int) -> List[Dict]:
        file = self.files[self.available_idx[index]]
        detection_batch: List[Dict] = []
        try:
            with H5File(os.path.join(self.memory_path, file)) as h5:
                memory = h5.read("memory_features")
                semmap_gt = h5.read("semmap_gt")
                proj_indices = h5.read("proj_indices")
        except Exception as e:                      # loader.py:205-208: a broken memory file degrades to an empty memory
            print(e)
            memory = np.zeros((1, 256))
            semmap_gt = None
            proj_indices = np.zeros((20, 480, 640, 1))

        if os.path.exists(self.semmap_path):
            from .snapshot import read_snapshot
            snap = read_snapshot(self.semmap_path, file)
            semmap_real, implicit_memory, observations = snap["semmap_real"], snap["implicit_memory"], snap["observations"]
        else:
            semmap_real, implicit_memory, observations = None, memory, None

        if self.clip_path:
            memory = np.insert(self.clip_embeddings, 0, np.zeros((1, 512)), axis=0)
            if self.memory_type == "map_gt":
                if semmap_real is not None:
                    proj_indices = semmap_real[proj_indices]
                else:
                    memory = memory[self.smnet_class_mapping]
                    proj_indices = semmap_gt[proj_indices]

        with H5File(os.path.join(self.data_path, file)) as h5:
            segmentation_data = h5.read("segmentation_data") if self.clip_path and self.memory_type == "semantic_gt" else None
            records = h5.read_strings("detection_data")
            for i in range(min(self.max_sequence_length, len(records))):
                file_name, gt_box, gt_class = parse_detection_data(records[i])
                keep = [k for k in range(len(gt_class)) if gt_class[k] in self.class_ids]
                gt_box = [gt_box[k] for k in keep]
                gt_class = [gt_class[k] for k in keep]
                if segmentation_data is not None:
                    proj_indices[i] = segmentation_data[i].reshape(segmentation_data.shape[1], segmentation_data.shape[2], 1)
                rgb_i = read_image_rgb(os.path.join(self.image_root, file_name))
                if self.test_type in ("default", "longterm"):
                    seq_id = int(file.split("_")[-1].split(".")[0])
                    mem_reset = seq_id == 0 and i == 0
                else:                                # episodic
                    mem_reset = i == 0
                rec = {"file_name": file_name, "sequence_name": file, "gt_boxes": np.array(gt_box), "gt_classes": np.array(gt_class),
                       "image": rgb_i, "proj_indices": proj_indices[i], "memory_reset": mem_reset}
                if self.memory_type in ("explicit_map", "implicit_memory"):
                    rec["memory_features"], rec["observations"] = implicit_memory, observations
                else:
                    rec["memory_features"], rec["observations"] = memory, None
                detection_batch.append(rec)
        return detection_batch


def map_mp3d_batch_to_coco(data: List[List[Dict]]) -> List[List[Dict]]:
    """Loader records -> the frame dicts the model takes (train_mp3d.py:452-507)."""
    detection_batch = []
    for sample in data:
        detection_sequence = []
        for r in sample:
            rgb = r["image"]
            d = {"file_name": r["file_name"], "sequence_name": r["sequence_name"], "height": rgb.shape[0], "width": rgb.shape[1]}
            inst = Instances((rgb.shape[0], rgb.shape[1]))
            inst.set("gt_boxes", Boxes(torch.as_tensor(np.asarray(r["gt_boxes"], dtype=np.float32).reshape(-1, 4))))
            inst.set("gt_classes", torch.as_tensor(np.asarray(r["gt_classes"], dtype=np.int64).reshape(-1)))
            d["instances"] = inst
            d["image"] = torch.from_numpy(np.array(rgb)).permute(2, 0, 1)       # a writable copy (PIL arrays are read-only)
            d["memory"] = r["memory_features"]
            d["proj_indices"] = r["proj_indices"]
            d["memory_reset"] = r["memory_reset"]
            d["observations"] = r["observations"]
            detection_sequence.append(d)
        detection_batch.append(detection_sequence)
    return detection_batch


class _Scene:
    def __init__(self, loader: SMNetDetectionLoader, seq_id: int, name: str, indices: List[int]):
        self.loader, self.seq_id, self.name, self.indices = loader, seq_id, name, indices

    def episodes(self) -> Iterator[List[Dict]]:
        for i in self.indices:
            yield map_mp3d_batch_to_coco(collate_smnet([self.loader[i]]))[0]


class Mp3dScenes:
    """The dataset grouped by scene (all episodes of a scene stay on one rank, in file order; SURVEY §8e: the reference's
    `InferenceSampler` would cut index ranges across scene boundaries)."""

    def __init__(self, loader: SMNetDetectionLoader):
        self.loader = loader
        groups: Dict[str, List[int]] = {}
        for i, f in enumerate(loader.files):
            groups.setdefault(episode_sort_key(f)[0], []).append(i)
        self.scenes = [_Scene(loader, sid, name, idx) for sid, (name, idx) in enumerate(groups.items())]

    def __len__(self) -> int:
        return len(self.scenes)

    def shard(self, rank: int, world: int) -> List[_Scene]:
        return [s for s in self.scenes if s.seq_id % world == rank]

    def episode_offsets(self) -> Dict[int, int]:
        """Global dataloader index of each scene's first episode (keeps the aggregate independent of the sharding)."""
        return {s.seq_id: s.indices[0] for s in self.scenes}
