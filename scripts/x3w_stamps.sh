#!/bin/bash
# Diagnostic builds with cycle stamps (compile-time hooks only, never part of the product library), into build/var/.
# Run this here on the CPU box, then on a GPU box (gpurun):
#   python scripts/x3w_stamps.py build/var/lib_x3w_stamps.so            split-f16 weight gradient: cycles per chunk / entry, workgroup spans
#   python scripts/x3w_stamps.py build/var/lib_wgf_stamps.so f32        fp32 weight gradient
#   python scripts/x3w_stamps.py build/var/lib_wgb_stamps.so bf16       bf16 weight gradient
#   python scripts/x3b_stamps.py build/var/lib_x3b_stamps.so            split-f16 backward: cycles per part of one tile
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
make -s
mkdir -p ../../build/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value"
one() {   # source, tag, macro, extra flags
  /opt/rocm/bin/hipcc $FLAGS $4 -D$3 -c $1.hip -o ../../build/var/$1_$2.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/var/lib_$2.so $(ls *.o | grep -v "^$1.o$") ../../build/var/$1_$2.o -ldl
  echo built build/var/lib_$2.so
}
one wgrad_f16x3 x3w_stamps X3W_STAMPS -fno-slp-vectorize
one wgrad wgf_stamps WGF_STAMPS ""
one wgrad_bf16 wgb_stamps WGB_STAMPS ""
one mlp_bwd_f16x3 x3b_stamps X3B_STAMPS -fno-slp-vectorize
