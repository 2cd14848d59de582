#!/usr/bin/env python
"""Per-level check of the head's backward pieces on the pyramid of a 128x160 image (rows 320 / 80 / 20 / 6 / 2): GroupNorm + ReLU
backward and the 3x3 dgrad conv against torch autograd, level by level.  Diagnostics."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from embodied_object_detection_amd import ops

dev = torch.device("cuda:0")
shapes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
off = [0]
for h, w in shapes:
    off.append(off[-1] + h * w)
g = torch.Generator().manual_seed(int(os.environ.get("SEED", "0")))
C = 256
x = torch.randn((off[-1], C), generator=g) * float(os.environ.get("XSCALE", "1.0")) + float(os.environ.get("XSHIFT", "0.0"))
dy = torch.randn((off[-1], C), generator=g)
gamma = torch.rand(C, generator=g) + 0.5
beta = torch.randn(C, generator=g) * 0.2
st = ops.groupnorm_workspace(off, dev)
y = ops.groupnorm_relu(x.to(dev), gamma.to(dev), beta.to(dev), off, C, st)
dx, dga, dbe = ops.groupnorm_relu_backward(x.to(dev), y, dy.to(dev), gamma.to(dev), off, C, st)
for l, (h, w) in enumerate(shapes):
    xl = x[off[l]:off[l + 1]].t().reshape(1, C, h, w).clone().requires_grad_()
    yl = F.relu(F.group_norm(xl, 32, gamma, beta, eps=1e-5))
    (yl * dy[off[l]:off[l + 1]].t().reshape(1, C, h, w)).sum().backward()
    ref = xl.grad.reshape(C, -1).t()
    fy = yl.detach().reshape(C, -1).t()
    print("level %d rows %3d: GN fwd err %.1e  bwd dx err %.1e (scale %.1e)" % (
        l, h * w, float((y[off[l]:off[l + 1]].cpu() - fy).abs().max()), float((dx[off[l]:off[l + 1]].cpu() - ref).abs().max()),
        float(ref.abs().max())))
wt = torch.randn((C, C, 3, 3), generator=g) * 0.03
conv = ops.Conv(wt, torch.zeros(C), pad=1, device=dev)
bwd = ops.ConvBackward(conv)
for l, (h, w) in enumerate(shapes):
    xin = torch.randn((1, h, w, C), generator=g)
    go = torch.randn((1, h, w, C), generator=g)
    o = bwd(xin.to(dev), None, go.to(dev))
    xr = xin.permute(0, 3, 1, 2).clone().requires_grad_()
    wr = wt.clone().requires_grad_()
    (F.conv2d(xr, wr, padding=1) * go.permute(0, 3, 1, 2)).sum().backward()
    rdx = xr.grad.permute(0, 2, 3, 1)
    rdw = wr.grad.permute(0, 2, 3, 1).reshape(C, -1)
    print("level %d rows %3d: dgrad err %.1e (scale %.1e)  wgrad err %.1e (scale %.1e)" % (
        l, h * w, float((o["dx"].cpu() - rdx).abs().max()), float(rdx.abs().max()), float((o["dw"].cpu() - rdw).abs().max()),
        float(rdw.abs().max())))
