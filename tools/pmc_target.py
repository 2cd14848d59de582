#!/usr/bin/env python
"""Target for rocprofv3 --pmc passes: the mask-head 3x3 GEMM (300 ROIs) in fp32-MFMA (tile 23) and bf16x3 (tile 54) form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = 300
x = torch.randn((R, 14, 14, 256), generator=g).to(dev)
conv = ops.Conv(torch.randn((256, 256, 3, 3), generator=g) * 0.05, torch.zeros(256), stride=1, pad=1, device=dev)
out = torch.empty((R, 14, 14, 256), device=dev)
for t in (23, 54):
    for _ in range(6):
        conv(x, R, 14, 14, relu=True, force_tile=t, out=out)
torch.cuda.synchronize()
