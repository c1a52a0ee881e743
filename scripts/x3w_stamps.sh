#!/bin/bash
# Diagnostic build of the split-f16 weight-gradient kernel with cycle stamps (-DX3W_STAMPS) into build/var/, to be run
# on a GPU box with: python scripts/x3w_stamps.py build/var/lib_x3w_stamps.so   (run this script here, on the CPU box)
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
make -s
mkdir -p ../../build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -fno-slp-vectorize -DX3W_STAMPS -c wgrad_f16x3.hip -o ../../build/var/wgrad_f16x3_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/var/lib_x3w_stamps.so $(ls *.o | grep -v '^wgrad_f16x3.o$') ../../build/var/wgrad_f16x3_stamps.o -ldl
echo built build/var/lib_x3w_stamps.so
