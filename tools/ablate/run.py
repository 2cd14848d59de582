#!/usr/bin/env python
"""Ablation of the implicit-GEMM main loop on the mask-head shape (diagnostic builds, wrong results by design)."""
import ctypes as C, os, subprocess, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, ROOT)
from embodied_object_detection_amd import _lib
from embodied_object_detection_amd.ops import pack_conv_weight
variants = {"full": [], "no_global": ["-DABL_NOGLOBAL"], "no_global_no_ldswrite": ["-DABL_NOGLOBAL", "-DABL_NOLDSW"], "no_barrier": ["-DABL_NOGLOBAL", "-DABL_NOLDSW", "-DABL_NOBARRIER"]}
dev = torch.device("cuda:0")
R = 256
x = torch.randn((R, 14, 14, 256), device=dev)
w, kpad = pack_conv_weight(torch.randn((256, 256, 3, 3)) * 0.05)
w = w.to(dev); y = torch.empty((R, 14, 14, 256), device=dev)
for name, flags in variants.items():
    so = os.path.join(HERE, f"abl_{name}.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-ffp-contract=on"] + flags + [os.path.join(HERE, "conv_ablate.hip"), "-o", so])
    lib = C.CDLL(so); lib.eod_conv2d.restype = C.c_int; lib.eod_conv2d.argtypes = [C.POINTER(_lib.EodConvDesc), C.c_void_p]
    for tile in (23, 22, 21, 3):
        d = _lib.EodConvDesc(); d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
        d.N, d.H, d.W, d.Cin, d.OH, d.OW, d.Cout, d.KH, d.KW, d.stride, d.pad, d.Kpad = R, 14, 14, 256, 14, 14, 256, 3, 3, 1, 1, kpad
        d.relu, d.out_scale, d.force_tile = 1, 1.0, tile
        s = torch.cuda.current_stream().cuda_stream
        assert lib.eod_conv2d(C.byref(d), s) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.eod_conv2d(C.byref(d), s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:24s} tile={tile:2d} {ms*1e3:8.1f} us {2.0*R*196*256*2304/ms/1e9:7.1f} TFLOP/s", flush=True)
