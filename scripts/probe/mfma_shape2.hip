// mfma_shape2.hip -- microprobe, second take: v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 on RANDOM data at the
// same 256 x 32 output per wave, one wave per SIMD, every CU busy.  mfma_shape.hip left the choice of registers to the
// compiler, which gave the 16x16x32 loop accumulator copies and overlapping source/destination ranges (28 cycles per MFMA
// on zeros instead of 16); here every MFMA is an asm statement that accumulates IN PLACE, so the loops are nothing but
// MFMAs.  Second mode: the A operand of every MFMA group is re-read from LDS by ds_read_b128 (what the render kernels do).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape2 mfma_shape2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF32(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// LDS mode: fragments are read 3 ahead into a ring of 4 register sets by asm ds_read_b128 with counted lgkmcnt waits (the
// compiler's own reads wait lgkmcnt(0) in front of every MFMA group and serialise read -> wait -> MFMA)
template <int OFF> __device__ __forceinline__ void dsr(f16x8& f, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
template <int J, int NF, int SHAPE, class ACC>
__device__ __forceinline__ void lds_steps(ACC& acc, f16x8 (&f)[4], const f16x8 (&b)[4], uint32_t addr) {
  if constexpr (J < NF) {
    dsr<((J + 3) % NF) * 1024>(f[(J + 3) & 3], addr);            // wraps into the next iteration's first fragments
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f[J & 3]));
    if constexpr (SHAPE == 0) {
      MF32(acc[J % 8], f[J & 3], b[J / 8]);                      // NF = 32: 4 k-steps x 8 tiles
    } else {
      MF16(acc[2 * (J % 16)], f[J & 3], b[2 * (J / 16)]);        // NF = 32: 2 k-steps x 16 tiles, two sample halves each
      MF16(acc[2 * (J % 16) + 1], f[J & 3], b[2 * (J / 16) + 1]);
    }
    lds_steps<J + 1, NF, SHAPE>(acc, f, b, addr);
  }
}

// SHAPE 0: 32x32x16, 8 accumulator tiles (256 features x 32 samples), per k-step of 16: 8 MFMAs, one A fragment each, one B
// SHAPE 1: 16x16x32, 16 feature tiles x 2 sample halves = 32 accumulators, per k-step of 32: 32 MFMAs, 16 A fragments, 2 B
// LDS: the A fragments come out of LDS (1 KiB per fragment per wave: ds_read_b128), else they stay in registers
template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256, 1) void k(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ f16x8 sm[LDS ? 64 * 64 : 1];      // 64 fragments of 1 KiB
  const int t = threadIdx.x + blockIdx.x * 256, lane = threadIdx.x & 63;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
  if constexpr (LDS) {
    for (int i = threadIdx.x; i < 64 * 64; i += 256) sm[i] = in[(i * 7 + blockIdx.x) & 0xffff];
    __syncthreads();
  }
  float s = 0.f;
  const uint32_t addr = (uint32_t)(size_t)sm + lane * 16;
  f16x8 f[4];
  if constexpr (LDS) { dsr<0>(f[0], addr); dsr<1024>(f[1], addr); dsr<2048>(f[2], addr); }
  if constexpr (SHAPE == 0) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if constexpr (LDS) {
        lds_steps<0, 32, 0>(acc, f, b, addr);
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < 8; ++i) MF32(acc[i], a[(i + kk) & 3], b[kk]);
      }
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if constexpr (LDS) {
        lds_steps<0, 32, 1>(acc, f, b, addr);
      } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 16; ++i) {      // one A fragment (16 features x 32 k) feeds both sample halves
            MF16(acc[2 * i], a[(i + kk) & 3], b[2 * kk]);
            MF16(acc[2 * i + 1], a[(i + kk) & 3], b[2 * kk + 1]);
          }
      }
    }
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  if constexpr (LDS) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); }
  out[t] = s;
}

template <int SHAPE, bool LDS>
static void run(const f16x8* in, float* out, int iters, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(256), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0) { sum += ms; if (ms < best) best = ms; }
  }
  // per iteration and wave: shape 0: 32 MFMAs x 32768 FLOP; shape 1: 64 MFMAs x 16384 FLOP -- the same 1 048 576
  const double flop = 1048576.0 * iters * 1024.0;
  printf("%-58s best %.3f ms  %.1f TFLOP/s   mean %.1f TFLOP/s\n", name, best, flop / best / 1e9, flop / (sum / 4) / 1e9);
}

int main() {
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  f16x8* in; float* out;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 65536 * 4);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int it = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0, false>(in, out, it, "32x32x16 f16, operands in registers, random data");
    run<1, false>(in, out, it, "16x16x32 f16, operands in registers, random data");
    run<0, true>(in, out, it, "32x32x16 f16, A from LDS (1 KiB per MFMA), random data");
    run<1, true>(in, out, it, "16x16x32 f16, A from LDS (1 KiB per 2 MFMAs), random data");
  }
  hipMemset(in, 0, h.size() * 2);
  run<0, false>(in, out, it, "32x32x16 f16, operands in registers, zeros");
  run<1, false>(in, out, it, "16x16x32 f16, operands in registers, zeros");
  run<0, true>(in, out, it, "32x32x16 f16, A from LDS, zeros");
  run<1, true>(in, out, it, "16x16x32 f16, A from LDS, zeros");
  return 0;
}
