#!/usr/bin/env python
"""Ablation of the bf16x3 split-MFMA main loop on the mask-head shape (diagnostic builds of the csrc/conv_*.hip files with -DABL_*;
wrong results by design).  Build the variants first (CPU container): python tools/ablate/run_bf16x3.py build"""
import ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "embodied_object_detection_amd", "csrc")
SRC = [os.path.join(CSRC, f) for f in ("conv_igemm.hip", "conv_fp32.hip", "conv_bf16x3.hip")]
variants = {"full": [], "no_global": ["-DABL_NOGLOBAL"], "no_split": ["-DABL_NOSPLIT"], "no_ldswrite": ["-DABL_NOLDSW"],
            "no_global_split": ["-DABL_NOGLOBAL", "-DABL_NOSPLIT"], "no_global_split_ldsw": ["-DABL_NOGLOBAL", "-DABL_NOSPLIT", "-DABL_NOLDSW"]}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for name, flags in variants.items():
        so = os.path.join(HERE, f"b3_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-ffp-contract=on"] + flags + SRC + ["-o", so])
    sys.exit(0)
import torch
from embodied_object_detection_amd import _lib
from embodied_object_detection_amd.ops import pack_conv_weight
dev = torch.device("cuda:0")
R = int(os.environ.get("ROIS", "300"))
x = torch.randn((R, 14, 14, 256), device=dev)
w, kpad = pack_conv_weight(torch.randn((256, 256, 3, 3)) * 0.05)
w = w.to(dev); y = torch.empty((R, 14, 14, 256), device=dev)
for rep in range(2):
  for name in variants:
    so = os.path.join(HERE, f"b3_{name}.so")
    lib = C.CDLL(so); lib.eod_conv2d.restype = C.c_int; lib.eod_conv2d.argtypes = [C.POINTER(_lib.EodConvDesc), C.c_void_p]
    for tile in (51, 54):
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
