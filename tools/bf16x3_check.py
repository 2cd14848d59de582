#!/usr/bin/env python
"""Accuracy (against an fp64 convolution) and speed of the bf16x3 split kernel next to the fp32-MFMA kernel (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from embodied_object_detection_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def accuracy(N, H, W, Cin, Cout, k, pad, tiles, scale_spread=False):
    x = torch.randn((N, Cin, H, W), generator=g)
    if scale_spread:                      # wide dynamic range: per-channel scales over 6 decades
        x = x * torch.logspace(-3, 3, Cin).view(1, Cin, 1, 1)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    b = torch.randn((Cout,), generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad)
    scale = ref.abs().mean().item()
    conv = ops.Conv(w, b, stride=1, pad=pad, device=dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = {}
    for t in tiles:
        y = conv(xd, N, H, W, force_tile=t, force_splitk=1).cpu().permute(0, 3, 1, 2).double()
        e = (y - ref).abs()
        out[t] = (e.max().item() / scale, e.mean().item() / scale)
    y32 = F.conv2d(x, w, b, padding=pad).double()
    e = (y32 - ref).abs()
    out["cpu_fp32"] = (e.max().item() / scale, e.mean().item() / scale)
    return out


def speed(name, N, H, W, Cin, Cout, k, stride, pad, tiles, iters=20):
    x = torch.randn((N, H, W, Cin), generator=g).to(dev)
    w = torch.randn((Cout, Cin, k, k), generator=g) * 0.05
    conv = ops.Conv(w, torch.zeros(Cout), stride=stride, pad=pad, device=dev)
    OH, OW = conv.out_hw(H, W)
    flops = 2.0 * N * OH * OW * Cout * Cin * k * k
    for t in tiles:
        out = conv(x, N, H, W, relu=True, force_tile=t)
        for _ in range(3):
            conv(x, N, H, W, relu=True, force_tile=t, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            conv(x, N, H, W, relu=True, force_tile=t, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{name:28s} tile={t:3d}  {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s (fp32-equivalent)", flush=True)


print("relative error vs fp64 (max, mean; normalised by mean |y|)")
for label, args in (("3x3 256->256 K=2304", (8, 14, 14, 256, 256, 3, 1)), ("1x1 2048->256", (1, 20, 20, 2048, 256, 1, 0)),
                    ("fc 12544->128", (64, 1, 1, 12544, 128, 1, 0))):
    for spread in (False, True):
        r = accuracy(*args, tiles=(23, 51, 52, 53), scale_spread=spread)
        print(label, "spread" if spread else "unit", {k: (f"{v[0]:.2e}", f"{v[1]:.2e}") for k, v in r.items()}, flush=True)

speed("mask_fcn 256 rois", 256, 14, 14, 256, 256, 3, 1, 1, tiles=(23, 51, 53, 54, 23, 51, 53, 54))
speed("mask_fcn 300 rois", 300, 14, 14, 256, 256, 3, 1, 1, tiles=(23, 51, 53, 54))
speed("tower 3x3 256 80x80", 1, 80, 80, 256, 256, 3, 1, 1, tiles=(0, 51, 52, 53, 54))
speed("l3 conv2 3x3 256 40x40", 1, 40, 40, 256, 256, 3, 1, 1, tiles=(0, 53))
speed("l4 conv3 1x1 512->2048", 1, 20, 20, 512, 2048, 1, 1, 0, tiles=(0, 53))
speed("fc1 256x12544->1024", 256, 1, 1, 12544, 1024, 1, 1, 0, tiles=(0, 53, 52))
speed("l1 conv2 3x3 64 160x160", 1, 160, 160, 64, 64, 3, 1, 1, tiles=(0, 53, 52))
