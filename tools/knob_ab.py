#!/usr/bin/env python
"""A/B of model attributes through the boundary (`model([episode])`, Instances materialised) in one process on the GPU box.

    python tools/knob_ab.py "" "early_memory_selection=True" "dedup_detection_masks=False;lazy_proposal_masks=False"

Each argument is a ';'-separated list of `attr=value` (attributes of the model; `roi_heads.x=v` / `backbone.x=v` reach the parts);
the empty string is the default configuration.  Every configuration is run twice (interleaved) on the same 60 resident frames."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import build_model, setup_cfg
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence

H = int(os.environ.get("AB_H", 640))
W = int(os.environ.get("AB_W", 640))
dev = torch.device("cuda:0")
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5,
                       "MODEL.MEMORY_CLS_SCORE_THRESH", 0.3, "MODEL.DEVICE", "cuda:0"])
sd = synthetic_state_dict(0)
N = 60
seq = SyntheticSequence(0, H=H, W=W, n_frames=N, map_w=200, map_h=200, cell=0.2)
frames = []
for i in range(N):
    f = seq.frame(i)
    f["image"] = f["image"].to(dev)
    f["proj_indices"] = torch.from_numpy(f["proj_indices"][..., 0]).to(dev)
    frames.append(f)


def apply(model, spec):
    for kv in filter(None, spec.split(";")):
        k, v = kv.split("=", 1)
        obj = model
        parts = k.strip().split(".")
        for p in parts[:-1]:
            obj = getattr(obj, p)
        if not hasattr(obj, parts[-1]):
            raise AttributeError(k)
        setattr(obj, parts[-1], eval(v))


def run(model):
    model([frames[:20]])
    torch.cuda.synchronize()
    out = []
    for _ in range(int(os.environ.get("AB_PASSES", 2))):      # consecutive 40-frame passes of the same model: does the rate hold?
        t = time.perf_counter()
        model([frames[20:40]])
        model([frames[40:60]])
        torch.cuda.synchronize()
        out.append(40 / (time.perf_counter() - t))
    return out


specs = sys.argv[1:] or [""]
res = {s: [] for s in specs}
for rep in range(int(os.environ.get("AB_REPS", 3))):
    for s in specs:
        m = build_model(cfg, sd)
        apply(m, s)
        res[s].append(run(m))
        del m
        torch.cuda.empty_cache()
for s in specs:
    print(f"{s or '(default)':70s} " + "  ".join("/".join(f"{v:.1f}" for v in r) for r in res[s]) + " frames/s", flush=True)
