#!/usr/bin/env python
"""Per-layer timing of one frame: HIP events around every implicit-GEMM launch (diagnostic, GPU box only)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import build_model, setup_cfg, ops
from embodied_object_detection_amd.checkpoint import synthetic_state_dict
from embodied_object_detection_amd.data.synthetic import SyntheticSequence

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 640
cfg = setup_cfg(None, ["MODEL.MEMORY_TYPE", "implicit_memory", "MODEL.MAP_FEAT_FUSION", "sum", "MODEL.MAP_FEATURE_WEIGHT", 5])
model = build_model(cfg, synthetic_state_dict(0))
seq = SyntheticSequence(0, H=H, W=W, n_frames=8)
frames = [seq.frame(i) for i in range(8)]
log = []
convs = []
def walk(o, seen):
    if id(o) in seen: return
    seen.add(id(o))
    if isinstance(o, ops.Conv): convs.append(o); return
    if isinstance(o, dict): [walk(v, seen) for v in o.values()]
    elif isinstance(o, (list, tuple)): [walk(v, seen) for v in o]
    elif hasattr(o, "__dict__"): [walk(v, seen) for v in vars(o).values()]
walk(model, set())
for i, f in enumerate(frames[:3]):
    model([[f]])
for c in convs: c.event_log = log
orig_call = ops.Conv.__call__
meta = []
def call(self, x, N, Hh, Ww, **k):
    out = orig_call(self, x, N, Hh, Ww, **k)
    OH, OW = self.out_hw(Hh, Ww)
    meta.append((self.name, N * OH * OW, self.Cout, self.Kpad))
    return out
ops.Conv.__call__ = call
nf = 0
for f in frames[3:]:
    model([[f]]); nf += 1
torch.cuda.synchronize()
agg = collections.OrderedDict()
for (e0, e1, c), (name, M, N, K) in zip(log, meta):
    rows = M if c is None else None
    d = agg.setdefault((name, M, N, K), [0.0, 0])
    d[0] += e0.elapsed_time(e1); d[1] += 1
tot = 0
print(f"{'layer':44s} {'M':>7s} {'N':>5s} {'K':>6s} {'us':>8s} {'TF(cap)':>8s} calls/frame")
for (name, M, N, K), (ms, n) in agg.items():
    us = ms / n * 1e3
    tot += ms / nf
    print(f"{name[-44:]:44s} {M:7d} {N:5d} {K:6d} {us:8.1f} {2.0*M*N*K/us/1e6:8.1f} {n/nf:.0f}")
print("sum of conv launches per frame (event-bracketed, includes launch gaps): %.3f ms" % tot)
