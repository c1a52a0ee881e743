// mfma_shape.hip -- microprobe: sustained rate of v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 on RANDOM data
// (operands in registers, one wave per SIMD, every CU busy): where the chip lowers its clock under the matrix pipe's
// load, the clock it holds depends on the shape (MI355X_MICROARCH.md).  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 32x32x16 (8 accumulators), 1: 16x16x32 (32 accumulators): the same 256 x 32 output per wave
__global__ __launch_bounds__(256, 1) void k(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = threadIdx.x + blockIdx.x * 256;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + kk) & 3], b[kk], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + kk) & 3], b[(i >> 4) + 2 * kk], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[t] = s;
}

template <int SHAPE>
static void run(const f16x8* in, float* out, int iters, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  // per iteration and wave: shape 0: 32 MFMAs x 32768 FLOP; shape 1: 64 MFMAs x 16384 FLOP -- the same 1 048 576
  const double flop = 1048576.0 * iters * 1024.0;
  printf("%s: %.3f ms  %.1f TFLOP/s\n", name, best, flop / best / 1e9);
}

int main() {
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  f16x8* in; float* out;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 65536 * 4);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(in, out, 20000, "v_mfma_f32_32x32x16_f16, random data");
    run<1>(in, out, 20000, "v_mfma_f32_16x16x32_f16, random data");
  }
  hipMemset(in, 0, h.size() * 2);
  run<0>(in, out, 20000, "v_mfma_f32_32x32x16_f16, zeros");
  run<1>(in, out, 20000, "v_mfma_f32_16x16x32_f16, zeros");
  return 0;
}
