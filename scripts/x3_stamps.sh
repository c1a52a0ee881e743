#!/bin/bash
# Cycle stamps of the split-f16 forward kernel: builds a DIAGNOSTIC copy of the library (build/libnerfhip_stamps.so,
# -DNERFHIP_DIAG_STAMPS=1, wrong results by construction) here on the CPU box; run the python part through gpurun:
#   bash scripts/x3_stamps.sh build   &&   gpurun -- python scripts/x3_stamps.py build/libnerfhip_stamps.so
set -e
cd "$(dirname "$0")/../nerf_amd/csrc"
make -s
mkdir -p ../../build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DNERFHIP_DIAG_STAMPS=1 -c mlp_fwd_f16x3.hip -o /tmp/x3_stamps.o
# every object of the product library except the forward kernel's, plus the instrumented one (as scripts/x3w_stamps.sh links:
# api.o references all kernels' launchers, so a hand-kept object list goes stale)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libnerfhip_stamps.so $(ls *.o | grep -v "^mlp_fwd_f16x3.o$") /tmp/x3_stamps.o -ldl
python3 -c "import ctypes, os; ctypes.CDLL('../../build/libnerfhip_stamps.so', mode=os.RTLD_NOW)"   # all symbols resolve
echo build/libnerfhip_stamps.so
