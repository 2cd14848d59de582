#!/usr/bin/env python
"""Stand-alone timing of the memory-read kernels on different index patterns: HIP events around 20 back-to-back launches behind a
1 ms spin kernel (the host has queued the batch before it starts), / 20."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_object_detection_amd import ops
from embodied_object_detection_amd.data.synthetic import SyntheticSequence

dev = torch.device("cuda:0")


def timed(fn, batch=20, reps=7):
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(2_000_000)
        a.record()
        for _i in range(batch):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / batch)
    return float(np.median(out[2:]))


for (H, W, mw, cell) in ((640, 640, 200, 0.2), (960, 960, 512, 0.08)):
    N = mw * mw
    seq = SyntheticSequence(0, H=H, W=W, n_frames=2, map_w=mw, map_h=mw, cell=cell)
    real = torch.from_numpy(seq.frame(1)["proj_indices"][..., 0]).to(dev)
    pats = {"synthetic": real, "constant": torch.full((H, W), 7, dtype=torch.int32, device=dev),
            "columns": (torch.arange(W, device=dev, dtype=torch.int32)[None, :] % N).expand(H, W).contiguous(),
            "distinct": (torch.arange(H * W, device=dev, dtype=torch.int32) % N).reshape(H, W).contiguous()}
    m16 = (torch.randn((N, 512), device=dev) * 10).half()
    out = torch.empty((ops.pooled_rows(H, W), 512), dtype=torch.float16, device=dev)
    for name, proj in pats.items():
        for order in (False, True):
            us = timed(lambda: ops.memory_gather_pool(m16, proj, H, W, out=out, torch_order=order))
            print(f"{H}x{W} gather_pool {name:10s} torch_order={int(order)} {us:8.1f} us")

# event-bracket overhead of a near-empty kernel, then normalise / project in isolation
from embodied_object_detection_amd import _lib
lib = _lib.load()
x = torch.zeros((64,), dtype=torch.float32, device=dev)
us = timed(lambda: lib.eod_fill_f32(x.data_ptr(), 0.0, 64, torch.cuda.current_stream().cuda_stream))
print(f"empty kernel (fill 64 floats) {us:8.1f} us")
for (H, W) in ((640, 640), (960, 960)):
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn((256, 512, 1, 1), generator=g) * 0.01 for _ in range(3)]
    bs = [torch.randn((256,), generator=g) * 0.01 for _ in range(3)]
    proj = ops.MemoryProjector(ws, bs, dev)
    pooled = (torch.randn((ops.pooled_rows(H, W), 512), device=dev)).half()
    feats = torch.randn((ops.pooled_rows(H, W) + 200, 256), device=dev)
    us = timed(lambda: proj(pooled, feats, H, W, 5.0, "sum"))
    print(f"{H}x{W} project_fuse {us:8.1f} us")
