// store_stream.hip -- microprobe: what rate do plain / non-temporal global stores reach when every CU streams a
// contiguous share (dword and dwordx4 stores, 256 or 512 workgroups of 256 threads)?  The ceiling for the kernels that
// write whole layers of activations / gradients.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WIDE, int NT>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, size_t floats_per_wg) {
  float* base = out + (size_t)blockIdx.x * floats_per_wg;
  const int t = threadIdx.x;
  if constexpr (WIDE) {
    f32x4 v = {1.f, 2.f, 3.f, (float)t};
    for (size_t i = (size_t)t * 4; i < floats_per_wg; i += 1024) {
      if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + i));
      else *reinterpret_cast<f32x4*>(base + i) = v;
    }
  } else {
    for (size_t i = t; i < floats_per_wg; i += 256) {
      if constexpr (NT) __builtin_nontemporal_store((float)t, base + i);
      else base[i] = (float)t;
    }
  }
}

template <int WIDE, int NT>
static void run(float* out, size_t bytes, int grid, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  const size_t fpw = bytes / 4 / grid / 1024 * 1024;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WIDE, NT>), dim3(grid), dim3(256), 0, 0, out, fpw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  printf("%s grid %d: %.3f ms  %.2f TB/s\n", name, grid, best, (double)fpw * 4 * grid / best / 1e9);
}

int main() {
  const size_t bytes = (size_t)8 << 30;
  float* out;
  if (hipMalloc(&out, bytes) != hipSuccess) return 1;
  for (int grid : {256, 512, 2048}) {
    run<0, 0>(out, bytes, grid, "dword stores          ");
    run<0, 1>(out, bytes, grid, "dword stores, nt      ");
    run<1, 0>(out, bytes, grid, "dwordx4 stores        ");
    run<1, 1>(out, bytes, grid, "dwordx4 stores, nt    ");
  }
  return 0;
}
