#!/bin/bash
# Cycle stamps of the split-f16 forward kernel: builds a DIAGNOSTIC copy of the library (build/libnerfhip_stamps.so,
# -DNERFHIP_DIAG_STAMPS=1, wrong results by construction) here on the CPU box; run the python part through gpurun:
#   bash scripts/x3_stamps.sh build   &&   gpurun -- python scripts/x3_stamps.py build/libnerfhip_stamps.so
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
make -s
mkdir -p ../../build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DNERFHIP_DIAG_STAMPS=1 -c mlp_fwd_f16x3.hip -o /tmp/x3_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libnerfhip_stamps.so api.o mlp_fwd.o mlp_bwd.o /tmp/x3_stamps.o mlp_fwd_bf16.o mlp_bwd_bf16.o wgrad_bf16.o wgrad.o ray_kernels.o pack.o -ldl
echo build/libnerfhip_stamps.so
