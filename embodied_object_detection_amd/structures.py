"""Minimal `Boxes` / `Instances` with the detectron2 surface the hot path and its callers use
(`Detic/train_mp3d.py:186-246` reads `output["instances"]`, `.pred_boxes.tensor`, `.scores`, `.pred_classes`)."""
from __future__ import annotations

from typing import Any, Dict, Tuple

import torch


class Boxes:
    def __init__(self, tensor: torch.Tensor):
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def __len__(self) -> int:
        return self.tensor.shape[0]

    def __getitem__(self, item) -> "Boxes":
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def to(self, *a, **k) -> "Boxes":
        return Boxes(self.tensor.to(*a, **k))

    def area(self) -> torch.Tensor:
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    @property
    def device(self):
        return self.tensor.device

    def __repr__(self):
        return f"Boxes({self.tensor})"


class Instances:
    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def __setattr__(self, name: str, val: Any) -> None:
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str) -> Any:
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name: str, value: Any) -> None:
        n = len(value)
        if len(self._fields):
            assert len(self) == n, f"Adding a field of length {n} to Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name: str) -> bool:
        return name in self._fields

    def remove(self, name: str) -> None:
        del self._fields[name]

    def get(self, name: str) -> Any:
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *a, **k) -> "Instances":
        ret = Instances(self._image_size)
        for key, v in self._fields.items():
            ret.set(key, v.to(*a, **k) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item) -> "Instances":
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self) -> int:
        for v in self._fields.values():
            return len(v)
        return 0

    def __repr__(self):
        return f"Instances(num={len(self)}, image_size={self._image_size}, fields={list(self._fields)})"
