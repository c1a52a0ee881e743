// mfma_shape_f32.hip -- microprobe: v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 on RANDOM data, in-place asm
// accumulation, operands in registers, one wave per SIMD, every CU busy (the fp32 kernels run at 2.23-2.33 GHz: does the
// other shape hold a higher clock, as it does for the f16 / bf16 shapes -- mfma_shape2.hip?)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_f32 mfma_shape_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF32(acc, a, b) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF16(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
template <int SHAPE>   // 0: 32x32x2 (8 accumulators: 256 x 32 output); 1: 16x16x4 (32 accumulators: the same output)
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = threadIdx.x + blockIdx.x * 256;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xffff]; b[i] = in[(t * 16 + 8 + i) & 0xffff]; }
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < 8; ++i) MF32(acc[i], a[(i + kk) & 7], b[kk]);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 32; ++i) MF16(acc[i], a[(i + kk) & 7], b[2 * kk + (i & 1)]);
    }
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[t] = s;
}
template <int SHAPE>
static void run(const float* in, float* out, int iters, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  // per iteration and wave: shape 0: 64 MFMAs x 4096 FLOP; shape 1: 128 MFMAs x 2048 FLOP -- the same 262 144
  const double flop = 262144.0 * iters * 1024.0;
  printf("%-44s best %.3f ms  %.1f TFLOP/s\n", name, best, flop / best / 1e9);
}
int main() {
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  float *in, *out;
  (void)hipMalloc(&in, h.size() * 4); (void)hipMalloc(&out, 65536 * 4);
  (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int it = 8000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(in, out, it, "32x32x2 f32, random data");
    run<1>(in, out, it, "16x16x4 f32, random data");
  }
  (void)hipMemset(in, 0, h.size() * 4);
  run<0>(in, out, it, "32x32x2 f32, zeros");
  run<1>(in, out, it, "16x16x4 f32, zeros");
  return 0;
}
