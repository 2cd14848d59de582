#!/usr/bin/env python
"""gather_pool time per frame of the synthetic sequence (the kernel's cost depends on how many distinct cells a 16x16 pixel quadrant
sees): 20 back-to-back launches between one pair of events, / 20."""
import os, sys
import numpy as np
import torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_object_detection_amd import ops
from embodied_object_detection_amd.data.synthetic import SyntheticSequence

dev = torch.device("cuda:0")
H = W = 640
seq = SyntheticSequence(int(sys.argv[1]) if len(sys.argv) > 1 else 0, H=H, W=W, n_frames=80)
m16 = (torch.randn((seq.n_cells, 512), device=dev) * 10).half()
out = torch.empty((ops.pooled_rows(H, W), 512), dtype=torch.float16, device=dev)
blocker = torch.empty((64 << 20,), dtype=torch.float32, device=dev)
for i in (0, 2, 5, 7, 10, 12, 17, 22, 30, 40, 50, 60, 70, 79):
    p = seq.frame(i)["proj_indices"][..., 0]
    q = p.reshape(H // 16, 16, W // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 256)
    nq = np.array([len(np.unique(x)) for x in q])
    proj = torch.from_numpy(p).to(dev)
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _f in range(5):
            blocker.fill_(0.0)
        a.record()
        for _k in range(20):
            ops.memory_gather_pool(m16, proj, H, W, out=out)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 20)
    print(f"frame {i:3d}: distinct cells per 16x16 quadrant mean {nq.mean():5.2f}, > 16 in {100 * (nq > 16).mean():4.1f} % of quadrants, "
          f"{len(np.unique(p)):5d} per frame; gather_pool {np.median(ts[1:]):6.1f} us")
