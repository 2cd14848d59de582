#!/bin/bash
# Variant of the library whose selection kernels write phase stamps (tools/select_stamps.py).  Needs the objects of a normal build.
set -e
P=/root/repo/embodied_object_detection_amd
mkdir -p /root/repo/tools/ablate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -ffp-contract=on -DEOD_STAMPS -c $P/csrc/select.hip -o /root/repo/tools/ablate/select_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/ablate/libeod_stamps.so $(ls $P/build/*.o | grep -v "select.o\|conv_glds\|conv_halo") /root/repo/tools/ablate/select_stamps.o
ls -la /root/repo/tools/ablate/libeod_stamps.so
