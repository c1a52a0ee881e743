// wgrad_f16x3.hip -- weight-gradient GEMM of a training step with dtype f16x3 (gfx950): the function of wgrad.hip
//   dW[i][j] = sum_m  A[i][m] * B[j][m]          A = dY^T (gradient rows), B = X^T (input feature rows)
// on the SAME fp32 operands in the SAME tile-major store, staged into the same swizzled LDS image by the same
// persistent work list -- but contracted on the f16 matrix pipe: every fragment a wave reads from LDS (four fp32
// samples of a row) is split in registers into two f16 halves (f16x3_common.cuh: v = hi + lo) and a 16-sample k-step
// takes three v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi) into the fp32 accumulators, 5.3 x fewer matrix-pipe
// cycles than v_mfma_f32_32x32x2_f32 spends on the product.  The kernel is then bound by its 20 GB operand stream.
// Scale: the gradient rows are 1e-4 .. 1e-12, so A is multiplied by a power of two G chosen from the largest |dY| of
// the product's gradient buffer (published by mlp_bwd_f16x3.hip, WgradItem::amax) that puts that element at 2^12;
// an element 2^-14 below the largest one keeps 22 bits, smaller ones are exact to 2^-37 of the largest -- a sum over
// a million samples dominated by the large terms does not see it.  The accumulators are multiplied by 1/G on their
// way out (exact).  Bias gradients stay fp32 row sums of the A fragments.
#include "f16x3_common.cuh"
#include <type_traits>
#include "launchers.h"

namespace nh {
namespace x3 {

constexpr int kWgRows = 256;                 // max rows per operand tile
constexpr int kWgChunkFloats = kWgRows * 32; // one operand, one chunk: 256 rows x 32 samples
constexpr int kWgBufBytes = 2 * kWgChunkFloats * 4;   // one buffer = A chunk | B chunk = 64 KiB

template <int IMM>
__device__ __forceinline__ void lds_read16(f32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int NT>
__device__ __forceinline__ void read_frags(f32x4 (&f)[NT], uint32_t addr) {   // tile i = rows +32i = +4096 B
  lds_read16<0>(f[0], addr);
  if constexpr (NT > 1) lds_read16<4096>(f[1], addr);
  if constexpr (NT > 2) { lds_read16<8192>(f[2], addr); lds_read16<12288>(f[3], addr); }
}
template <int NT>
__device__ __forceinline__ void tie_frags(f32x4 (&f)[NT]) {
  asm volatile("" : "+v"(f[0]));
  if constexpr (NT > 1) asm volatile("" : "+v"(f[1]));
  if constexpr (NT > 2) { asm volatile("" : "+v"(f[2])); asm volatile("" : "+v"(f[3])); }
}
__device__ __forceinline__ void gatomic_add(float* p, float v) {
  __hip_atomic_fetch_add((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N)); }

// four fp32 samples -> dwords 2*HALF, 2*HALF+1 of an operand pair: hi = f16(v * k), lo = f16(v * k - hi)
template <int HALF>
__device__ __forceinline__ void split4_by(Pair& p, const f32x4& v, float k) {
  uint32_t h0, h1, l0, l1;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h0) : "v"(v[0]), "s"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h0) : "v"(v[1]), "s"(k));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h1) : "v"(v[2]), "s"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h1) : "v"(v[3]), "s"(k));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v[0]), "s"(k), "v"(h0));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v[1]), "s"(k), "v"(h0));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v[2]), "s"(k), "v"(h1));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v[3]), "s"(k), "v"(h1));
  p.hi[2 * HALF] = h0; p.hi[2 * HALF + 1] = h1;
  p.lo[2 * HALF] = l0; p.lo[2 * HALF + 1] = l1;
}
template <int HALF>
__device__ __forceinline__ void split4(Pair& p, const f32x4& v) {   // unscaled
  split_into(p, 2 * HALF, v[0], v[1]);
  split_into(p, 2 * HALF + 1, v[2], v[3]);
}

// One workgroup = one work entry; wave (wa, wb) owns A tiles [AT*wa, +AT) x B tiles [BT*wb, +BT).
template <int AT, int BT>
__device__ __forceinline__ void wgrad_body(const WgradItem& it, float* smem, int wave, int lane) {
  static_assert((AT == 1 || AT == 2 || AT == 4) && (BT == 1 || BT == 2 || BT == 4), "");
  const int li = lane & 31, h = lane >> 5;
  const int wa = wave / it.wb_n, wb = wave - wa * it.wb_n;
  const int a_row0 = 32 * AT * wa, b_row0 = 32 * BT * wb;
  const int c0 = it.c0, c1 = it.c1;   // this work entry's slice of the sample axis, in 32-sample chunks

  // scale of the gradient operand: its largest element (times 2^amax_shift for sums of elements) -> [2^12, 2^13)
  float G = 1.f, Ginv = 1.f;
  {
    const float am = it.amax ? *it.amax : 0.f;
    if (am > 0.f) {
      int e = 12 - ((int)((__builtin_bit_cast(uint32_t, am) >> 23) & 0xffu) - 127) - it.amax_shift;
      e = __builtin_amdgcn_readfirstlane(min(max(e, -100), 100));
      G = __builtin_bit_cast(float, (uint32_t)(e + 127) << 23);
      Ginv = __builtin_bit_cast(float, (uint32_t)(127 - e) << 23);
    }
  }

  f32x16 acc[AT][BT];
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int j = 0; j < BT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias gradient = row sums of A (fp32, from the fragments as they come out of LDS), by the waves with wb == 0
  float bs[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) bs[i] = 0.f;
  const bool do_bias = it.db != nullptr && wb == 0;

  // per-lane LDS byte addresses: row (row0 + li) of the A / B image + swizzled slot of k-group g
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem;
  const uint32_t sw = (uint32_t)((li >> 1) & 7);
  const uint32_t laneA = lds0 + (uint32_t)(a_row0 + li) * 128u;
  const uint32_t laneB = lds0 + (uint32_t)kWgChunkFloats * 4u + (uint32_t)(b_row0 + li) * 128u;
  uint32_t xoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xoff[g] = 16u * ((uint32_t)(2 * g + h) ^ sw);

  // Staging as in wgrad.hip: wave-instruction u of an operand covers rows 8u..8u+7; this wave issues u = wave + 4t.
  const int64_t a_step = (int64_t)it.a_stride * 32, b_step = (int64_t)it.b_stride * 32;   // floats per chunk
  const int nA8 = (it.a_rows + 7) >> 3, nB8 = (it.b_rows + 7) >> 3;
  uint32_t offA[8], offB[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int r = 8 * (wave + 4 * t) + (lane >> 3), sl = (lane & 7) ^ ((r >> 1) & 7);
    offA[t] = (uint32_t)((min(r, it.a_rows - 1) * 32 + 4 * sl) * 4);   // rows past the end repeat the last row (never stored)
    offB[t] = (uint32_t)((min(r, it.b_rows - 1) * 32 + 4 * sl) * 4);
  }
  auto stage_t = [&](int c, int buf, int t) __attribute__((always_inline)) {   // pieces u = wave + 4t of both operands of chunk c
    float* dst = smem + buf * 2 * kWgChunkFloats;
    const char* ga = reinterpret_cast<const char*>(it.A + c * a_step);
    const char* gb = reinterpret_cast<const char*>(it.B + c * b_step);
    const int u = wave + 4 * t;
    const uint32_t oa = t == 0 ? offA[0] : t == 1 ? offA[1] : t == 2 ? offA[2] : t == 3 ? offA[3] : t == 4 ? offA[4] : t == 5 ? offA[5] : t == 6 ? offA[6] : offA[7];
    const uint32_t ob = t == 0 ? offB[0] : t == 1 ? offB[1] : t == 2 ? offB[2] : t == 3 ? offB[3] : t == 4 ? offB[4] : t == 5 ? offB[5] : t == 6 ? offB[6] : offB[7];
    if (u < nA8) __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(ga + oa), NERFHIP_LPTR(dst + u * 256), 16, 0, 0);
    if (u < nB8) __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(gb + ob), NERFHIP_LPTR(dst + kWgChunkFloats + u * 256), 16, 0, 0);
  };
  auto stage = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 8; ++t) stage_t(c, buf, t);
  };
  stage(c0, 0);
  if (c0 + 1 < c1) stage(c0 + 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ONE set of fragment registers: a k-group is converted into its half of the operand pairs as soon as it has
  // landed, which frees the set for the next group's reads; those are issued in front of the k-step's MFMAs where there
  // are any (odd groups), else their latency is exposed (twice per chunk, ~200 cycles of the ~6000 a chunk's bytes take
  // to arrive).  A second set costs 32 registers that the 4 x 4 shape does not have: with it the compiler parks
  // freshly loaded fragment registers in AGPRs while the loads are still in flight (scripts/check_inflight.py).
  f32x4 fa[AT], fb[BT];
  Pair opA[AT], opB[BT];   // the 16-sample k-step being assembled: k-groups 2s (dwords 0, 1) and 2s+1 (dwords 2, 3)
  auto reads = [&](uint32_t bufoff, int g) __attribute__((always_inline)) {
    read_frags<AT>(fa, laneA + bufoff + xoff[g]); read_frags<BT>(fb, laneB + bufoff + xoff[g]);
  };
  auto landed = [&]() __attribute__((always_inline)) { wait_lgkm<0>(); tie_frags<AT>(fa); tie_frags<BT>(fb); };
  // the landed k-group -> half HALF of the operand pairs
  auto convert = [&](auto half) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half)::value;
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < AT; ++i) bs[i] += (fa[i][0] + fa[i][1]) + (fa[i][2] + fa[i][3]);
    }
#pragma unroll
    for (int i = 0; i < AT; ++i) split4_by<HALF>(opA[i], fa[i], G);
#pragma unroll
    for (int j = 0; j < BT; ++j) split4<HALF>(opB[j], fb[j]);
    // the fragment registers are free from here on: keep the conversions in front of the next reads
    asm volatile("" : "+v"(opA[0].hi), "+v"(opB[0].hi));
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfmas = [&]() __attribute__((always_inline)) {   // one 16-sample k-step: 3 AT BT MFMAs
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(opA[i].hi), as_f16x8(opB[j].hi), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(opA[i].hi), as_f16x8(opB[j].lo), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(opA[i].lo), as_f16x8(opB[j].hi), acc[i][j], 0, 0, 0);
      }
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  auto burst = [&](int part, bool go, int pc, int pbuf) __attribute__((always_inline)) {   // staging pieces of chunk pc
    if (go) {
      if (part == 0) { stage_t(pc, pbuf, 0); stage_t(pc, pbuf, 1); }
      if (part == 1) { stage_t(pc, pbuf, 2); stage_t(pc, pbuf, 3); stage_t(pc, pbuf, 4); }
      if (part == 2) { stage_t(pc, pbuf, 5); stage_t(pc, pbuf, 6); stage_t(pc, pbuf, 7); }
    }
  };
  // One chunk: k-groups 0..3 = two k-steps.  With a chunk to follow, the hand-over to its buffer (DMA landed for
  // everyone, first fragment reads) sits in front of the last k-step's MFMAs.
  auto chunk = [&](int c, bool has_next) __attribute__((always_inline)) {
    const uint32_t cur = (uint32_t)((c - c0) & 1) * (uint32_t)kWgBufBytes;
    const bool more = c > c0 && c + 1 < c1;   // chunk c+1's parts 1 and 2 go into the buffer chunk c-1 used
    const int nbuf = (c + 1 - c0) & 1;
    landed(); convert(H0{});                  // group 0
    reads(cur, 1);
    burst(1, more, c + 1, nbuf);
    landed(); convert(H1{});                  // group 1
    reads(cur, 2);
    burst(2, more, c + 1, nbuf);
    mfmas();                                  // k-step 0, over the reads of group 2
    landed(); convert(H0{});                  // group 2
    reads(cur, 3);
    landed(); convert(H1{});                  // group 3: every wave is done READING this buffer
    if (has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk c+1 must have landed for all
      __syncthreads();
      burst(0, c + 2 < c1, c + 2, (c - c0) & 1);
      reads(cur ^ (uint32_t)kWgBufBytes, 0);
    }
    mfmas();                                  // k-step 1, over the first reads of the next chunk
  };
  reads(0u, 0);
  for (int c = c0; c < c1 - 1; ++c) chunk(c, true);
  chunk(c1 - 1, false);   // peeled: no fragment read is in flight when the loop is left
  // combine the sample-axis slices: fp32 atomics, 128 contiguous bytes per lane half per instruction
#pragma unroll
  for (int i = 0; i < AT; ++i) {
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      const int col = b_row0 + 32 * j + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = a_row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < it.a_rows && col < it.b_rows) gatomic_add(it.dW + (int64_t)row * it.ldo + it.col0 + col, acc[i][j][r] * Ginv);
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      const float v = bs[i] + __shfl_xor(bs[i], 32);
      const int row = a_row0 + 32 * i + li;
      if (h == 0 && row < it.a_rows) gatomic_add(it.db + row, v);
    }
  }
}

// Persistent workgroup b works through the entries first[b] .. first[b+1]-1 (host: train_impl.inc).
__global__ __launch_bounds__(256, 1) void wgrad_f16x3_kernel(const WgradItem* __restrict__ items,
                                                             const int32_t* __restrict__ first) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 buffers x (A chunk | B chunk)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int e0 = first[blockIdx.x], e1 = first[blockIdx.x + 1];
  for (int e = e0; e < e1; ++e) {
    const WgradItem it = items[e];
    if (it.c0 >= it.c1) continue;
    switch (it.shape) {   // (tiles per wave along A) * 16 + (tiles per wave along B), chosen by the host
      case 0x44: wgrad_body<4, 4>(it, smem, wave, lane); break;
      case 0x42: wgrad_body<4, 2>(it, smem, wave, lane); break;
      case 0x41: wgrad_body<4, 1>(it, smem, wave, lane); break;
      case 0x24: wgrad_body<2, 4>(it, smem, wave, lane); break;
      case 0x22: wgrad_body<2, 2>(it, smem, wave, lane); break;
      case 0x21: wgrad_body<2, 1>(it, smem, wave, lane); break;
      case 0x14: wgrad_body<1, 4>(it, smem, wave, lane); break;
      case 0x12: wgrad_body<1, 2>(it, smem, wave, lane); break;
      default:   wgrad_body<1, 1>(it, smem, wave, lane); break;
    }
    __syncthreads();   // every wave is done reading the LDS buffers before the next entry stages into them
  }
}

}  // namespace x3

// once per device, with that device current (nerfhip_create): allow the kernel its 128 KiB of dynamic LDS
hipError_t prepare_wgrad_f16x3() {
  return hipFuncSetAttribute((const void*)x3::wgrad_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)((size_t)4 * x3::kWgChunkFloats * sizeof(float)));
}

hipError_t launch_wgrad_f16x3(const WgradItem* d_items, const int32_t* d_first, int ngroups, hipStream_t s) {
  if (ngroups <= 0) return hipSuccess;
  const size_t lds = (size_t)4 * x3::kWgChunkFloats * sizeof(float);  // 128 KiB
  hipLaunchKernelGGL(x3::wgrad_f16x3_kernel, dim3(ngroups), dim3(256), lds, s, d_items, d_first);
  return hipGetLastError();
}

}  // namespace nh
