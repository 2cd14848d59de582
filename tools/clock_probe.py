#!/usr/bin/env python
"""What engine clock does the chip hold under sustained fp32-MFMA load?  Runs the dominant kernel (mask_fcn 3x3 implicit GEMM, 300
ROIs) back to back for a few seconds while a thread samples the current sclk from sysfs (`pp_dpm_sclk`, the line marked `*`) and
`rocm-smi --showclocks`, and prints the achieved TFLOP/s beside the sampled clocks: the fp32 MFMA peak of MI355X_MICROARCH.md (157.3
TFLOP/s) is quoted at the boost clock; the attainable ceiling scales with the clock the chip really sustains.

    python tools/clock_probe.py [seconds]"""
import glob
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_object_detection_amd import ops

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rois = 300
x = torch.randn((rois, 14, 14, 256), generator=g).to(dev)
conv = ops.Conv(torch.randn((256, 256, 3, 3), generator=g) * 0.05, torch.zeros(256), pad=1, device=dev)
out = conv(x, rois, 14, 14, relu=True)
torch.cuda.synchronize()
flop = 2.0 * rois * 196 * 256 * 2304

samples = []
stop = False


def sysfs_sclk():
    vals = []
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(p):
                if "*" in line:
                    m = re.search(r"(\d+)\s*Mhz", line, re.I)
                    if m:
                        vals.append(int(m.group(1)))
        except OSError:
            pass
    return vals


def smi_sclk():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        return [int(v) for v in re.findall(r"sclk clock level[^\n]*?\((\d+)Mhz\)", o)]
    except Exception:
        return []


def sampler():
    while not stop:
        samples.append((time.perf_counter(), sysfs_sclk(), smi_sclk()))
        time.sleep(0.2)


print("idle:", sysfs_sclk(), smi_sclk(), flush=True)
th = threading.Thread(target=sampler)
th.start()
t0 = time.perf_counter()
rates = []
while time.perf_counter() - t0 < SECONDS:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        conv(x, rois, 14, 14, relu=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    rates.append((time.perf_counter() - t0, flop * 200 / (e0.elapsed_time(e1) * 1e-3) / 1e12))
stop = True
th.join()
for t, r in rates:
    print(f"t = {t:5.2f} s   {r:6.1f} TFLOP/s  ({r / 157.3:.3f} of 157.3)", flush=True)
for t, a, b in samples:
    print(f"t = {t - t0:5.2f} s   sysfs sclk {a}   rocm-smi sclk {b}", flush=True)
