"""Single-frame predictor of the robot demo (`EmbodiedPredictor`, `Detic/detic/predictor.py:389-439`).

Same call shape: `pred(data)` with `data = {"image": HWC uint8, "memory", "proj_indices", "memory_reset", "sequence_name"}` ->
`{"instances": Instances}`; like the reference it reverses the channel order when `INPUT.FORMAT == "RGB"` (`:421-423`: the
detectron2 predictor convention of BGR inputs), keeps `height` / `width` of the original image and runs
`model([[inputs]])[0]`.  `ResizeShortestEdge([480, 480], INPUT.MAX_SIZE_TEST)` (`:402-404`) is the identity for the 480 x 640
frames of this path; any other size is refused, because the reference resizes the image but not `proj_indices` and then fails
inside the memory fusion (SURVEY §8 notation).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch


class EmbodiedPredictor:
    def __init__(self, cfg, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        from .. import build_model
        self.cfg = cfg
        self.model = build_model(cfg, state_dict)          # loads cfg.MODEL.WEIGHTS like DetectionCheckpointer (:399-400)
        self.input_format = cfg.INPUT.FORMAT
        assert self.input_format in ("RGB", "BGR"), self.input_format
        self.max_size = int(cfg.INPUT.MAX_SIZE_TEST)

    def _resized_hw(self, h: int, w: int):
        """`ResizeShortestEdge([480, 480], max_size).get_output_shape`."""
        scale = 480.0 / min(h, w)
        nh, nw = (480.0, scale * w) if h < w else (scale * h, 480.0)
        if max(nh, nw) > self.max_size:
            s = self.max_size / max(nh, nw)
            nh, nw = nh * s, nw * s
        return int(nh + 0.5), int(nw + 0.5)

    def __call__(self, data: Dict) -> Dict:
        original_image = np.asarray(data["image"])
        if self.input_format == "RGB":
            original_image = original_image[:, :, ::-1]
        height, width = original_image.shape[:2]
        if self._resized_hw(height, width) != (height, width):
            raise ValueError(f"image {height}x{width}: the predictor's resize would change the image but not proj_indices "
                             "(the reference path only works for frames whose shortest edge is 480)")
        image = torch.from_numpy(np.ascontiguousarray(original_image)).permute(2, 0, 1)
        inputs = {"image": image, "height": height, "width": width, "memory": data["memory"], "proj_indices": data["proj_indices"],
                  "memory_reset": data["memory_reset"], "sequence_name": data["sequence_name"]}
        return self.model([[inputs]])[0]
