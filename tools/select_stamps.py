#!/usr/bin/env python
"""Where the time goes inside the two single-workgroup selection kernels (`cn_merge_nms_kernel`, `det_select_kernel`): runs a few
frames on ONE stream against a build of the library with -DEOD_STAMPS (tools/ablate/libeod_stamps.so: thread 0 writes the 100 MHz
wall clock at named points) and prints the phases in microseconds.  Diagnostics only.

    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=on -DEOD_STAMPS -c csrc/select.hip -o tools/ablate/select_stamps.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ablate/libeod_stamps.so <the other objects of build/> tools/ablate/select_stamps.o
    python tools/select_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embodied_object_detection_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "tools", "ablate", "libeod_stamps.so")
import numpy as np
import torch
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence

lib = _lib.load()
lib.eod_debug_read_stamps.restype = ctypes.c_int
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5,
                       "MODEL.MEMORY_CLS_SCORE_THRESH", 0.3])
model = build_model(cfg, synthetic_state_dict(0))
model.overlap_branches = False
model.prefetch_trunk = False
seq = SyntheticSequence(0, H=640, W=640, n_frames=12)
rows = []
for i in range(12):
    f = seq.frame(i)
    model([[f]])
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.eod_debug_read_stamps(buf) == 0
    rows.append(np.array(list(buf), dtype=np.int64))
a = np.stack(rows[4:])                 # steady state
us = lambda i, j: np.median((a[:, j] - a[:, i]) / 100.0)
print("cn_merge_nms_kernel: load keys %.1f | prefilter + sort %.1f | decode %.1f | nms walk %.1f (%d chunks) | outputs %.1f | total %.1f us"
      % (us(0, 1), us(1, 2), us(2, 3), us(3, 5), int(np.median(a[:, 6])), us(5, 4), us(0, 4)))
for name, sb in (("det_select_kernel (memory selection, topk 100)", 8), ("det_select_kernel (detections, topk 300)", 24)):
    print("%s: rows+scores %.1f | IoU matrix %.1f | histogram+cut %.1f | batch 0 compaction+sort %.1f | batch 0 walk %.1f | "
          "outputs %.1f | total %.1f us   (candidates %d, kept %d)"
          % (name, us(sb, sb + 1), us(sb + 1, sb + 2), us(sb + 2, sb + 3), us(sb + 3, sb + 4), us(sb + 4, sb + 5), us(sb + 5, sb + 8),
             us(sb, sb + 8), int(np.median(a[:, sb + 9])), int(np.median(a[:, sb + 10]))))
