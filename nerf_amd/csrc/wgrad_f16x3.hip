// wgrad_f16x3.hip -- weight-gradient GEMM of a training step with dtype f16x3 (gfx950): the function of wgrad.hip
//   dW[i][j] = sum_m  A[i][m] * B[j][m]          A = dY^T (gradient rows), B = X^T (input feature rows)
// on the SAME fp32 operands in the SAME tile-major store, staged into the same swizzled LDS image by the same
// persistent work list -- but contracted on the f16 matrix pipe: every fragment a wave reads from LDS (four fp32
// samples of a row) is split in registers into two f16 halves (f16x3_common.hip.h: v = hi + lo) and a 16-sample k-step
// takes three v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi) into the fp32 accumulators, 5.3 x fewer matrix-pipe
// cycles than v_mfma_f32_32x32x2_f32 spends on the product.  The kernel is then bound by its 20 GB operand stream.
// Scale: the gradient rows are 1e-4 .. 1e-12, so A is multiplied by a power of two G chosen from the largest |dY| of
// the product's gradient buffer (published by mlp_bwd_f16x3.hip, WgradItem::amax) that puts that element at 2^12;
// an element 2^-14 below the largest one keeps 22 bits, smaller ones are exact to 2^-37 of the largest -- a sum over
// a million samples dominated by the large terms does not see it.  The accumulators are multiplied by 1/G on their
// way out (exact).  Bias gradients stay fp32 row sums of the A fragments.
#include "f16x3_common.hip.h"
#include <type_traits>
#include "launchers.h"

namespace nh {
namespace x3 {

constexpr int kWgRows = 256;                 // max rows per operand tile
constexpr int kWgChunkFloats = kWgRows * 32; // one operand, one chunk: 256 rows x 32 samples
constexpr int kWgSlotBytes = kWgChunkFloats * 4;      // one ring slot = one operand of one chunk = 32 KiB
constexpr int kWgSlots = 5;                           // the whole 160 KiB of LDS

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
// s_waitcnt vmcnt(n) for a wave-uniform n in 0..8 (the count is an immediate)
__device__ __forceinline__ void wait_vm_upto8(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}
__device__ __forceinline__ int ring_next(int s) { return s + 2 >= kWgSlots ? s + 2 - kWgSlots : s + 2; }   // slot of q + 2
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N)); }

// four fp32 samples -> dwords 2*HALF, 2*HALF+1 of an operand pair: hi = f16(v * k), lo = f16(v * k - hi); in two
// parts of four instructions each, so that they can be issued between MFMAs a few at a time
template <int HALF>
__device__ __forceinline__ void split4_by_hi(Pair& p, const f32x4& v, float k) {
  uint32_t h0, h1;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h0) : "v"(v[0]), "s"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h0) : "v"(v[1]), "s"(k));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h1) : "v"(v[2]), "s"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h1) : "v"(v[3]), "s"(k));
  p.hi[2 * HALF] = h0; p.hi[2 * HALF + 1] = h1;
}
template <int HALF>
__device__ __forceinline__ void split4_by_lo(Pair& p, const f32x4& v, float k) {
  const uint32_t h0 = p.hi[2 * HALF], h1 = p.hi[2 * HALF + 1];
  uint32_t l0, l1;
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v[0]), "s"(k), "v"(h0));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v[1]), "s"(k), "v"(h0));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v[2]), "s"(k), "v"(h1));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v[3]), "s"(k), "v"(h1));
  p.lo[2 * HALF] = l0; p.lo[2 * HALF + 1] = l1;
}
// the same unscaled (f16x3_common.hip.h: split_into)
template <int HALF>
__device__ __forceinline__ void split4_hi(Pair& p, const f32x4& v) {
  const f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
  p.hi[2 * HALF] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v01, f16x2));
  p.hi[2 * HALF + 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v23, f16x2));
}
template <int HALF>
__device__ __forceinline__ void split4_lo(Pair& p, const f32x4& v) {
  const uint32_t h0 = p.hi[2 * HALF], h1 = p.hi[2 * HALF + 1];
  uint32_t l0, l1;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h0), "v"(v[0]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(h0), "v"(v[1]));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h1), "v"(v[2]));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(h1), "v"(v[3]));
  p.lo[2 * HALF] = l0; p.lo[2 * HALF + 1] = l1;
}
template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// One workgroup = one work entry; wave (wa, wb) owns A tiles [AT*wa, +AT) x B tiles [BT*wb, +BT).
// XR: the entry carries an extra vector product (WgradItem::x_row)
template <int AT, int BT, bool XR = false>
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
  const bool do_bias = it.db != nullptr && wb == 0;   // every wave sums (no branch among the MFMAs); these flush

  // per-lane LDS byte addresses: row (row0 + li) of the A / B image + swizzled slot of k-group g
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem;
  const uint32_t sw = (uint32_t)((li >> 1) & 7);
  const uint32_t laneA = lds0 + (uint32_t)(a_row0 + li) * 128u;
  const uint32_t laneB = lds0 + (uint32_t)(b_row0 + li) * 128u;
  uint32_t xoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xoff[g] = 16u * ((uint32_t)(2 * g + h) ^ sw);

  // Staging: wave-instruction u of an operand covers rows 8u..8u+7 (image as in wgrad.hip); this wave issues
  // u = wave + 4t.  The LDS is a ring of FIVE 32-KiB slots, each one operand of one chunk: operand-chunk
  // q = 2j + (0: A, 1: B) of the entry's j-th chunk lives in slot q % 5.  While chunk j is used, chunk j+1 has landed or
  // is landing and A of chunk j+2 is being requested into the slot B(j-1) left; the hand-over barrier in the middle of
  // chunk j frees A(j)'s slot for B(j+2).  The wait in front of that barrier leaves A(j+2) -- the youngest DMAs of
  // this wave -- in flight (a raw s_barrier: __syncthreads() would put vmcnt(0) in front of it).
  const int64_t a_step = (int64_t)it.a_stride * 32, b_step = (int64_t)it.b_stride * 32;   // floats per chunk
  const int a_stage = XR ? max(it.a_rows, it.x_row + 1) : it.a_rows;   // rows of A's buffer that are staged
  const int nA8 = (a_stage + 7) >> 3, nB8 = (it.b_rows + 7) >> 3;
  const int nPA = __builtin_amdgcn_readfirstlane(min(max((nA8 - wave + 3) >> 2, 0), 8));   // this wave's pieces of an A chunk
  uint32_t offA[8], offB[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int r = 8 * (wave + 4 * t) + (lane >> 3), sl = (lane & 7) ^ ((r >> 1) & 7);
    offA[t] = (uint32_t)((min(r, a_stage - 1) * 32 + 4 * sl) * 4);   // rows past the end repeat the last row (never stored)
    offB[t] = (uint32_t)((min(r, it.b_rows - 1) * 32 + 4 * sl) * 4);
  }
  // piece T of operand OP (0: A, 1: B) of chunk c into ring slot `slot`
  auto stage1 = [&](auto op, auto t_, int c, int slot) __attribute__((always_inline)) {
    constexpr int OP = decltype(op)::value, T = decltype(t_)::value;
    float* dst = smem + slot * kWgChunkFloats;
    const char* g = OP == 0 ? reinterpret_cast<const char*>(it.A + c * a_step) : reinterpret_cast<const char*>(it.B + c * b_step);
    const int u = wave + 4 * T;
    if (u < (OP == 0 ? nA8 : nB8))
      __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(g + (OP == 0 ? offA[T] : offB[T])), NERFHIP_LPTR(dst + u * 256), 16, 0, 0);
  };
  auto stage_all = [&](auto op, int c, int slot) __attribute__((always_inline)) {
    static_for<8>([&](auto t_) __attribute__((always_inline)) { stage1(op, t_, c, slot); });
  };
  const int n = c1 - c0;
  stage_all(IC<0>{}, c0, 0);
  stage_all(IC<1>{}, c0, 1);
  if (n > 1) { stage_all(IC<0>{}, c0 + 1, 2); stage_all(IC<1>{}, c0 + 1, 3); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // The k-step being multiplied and the k-step being assembled: two sets of operand pairs.  A k-group (8 samples:
  // one 16-byte fragment per tile) comes out of LDS into ONE set of fragment registers and is converted into its half
  // of the pairs a few instructions at a time BETWEEN the MFMAs of the running k-step (back to back the conversions
  // of a chunk take longer than its MFMAs: 7.1 k cycles per chunk against 3.1 k of matrix pipe, measured).
  constexpr int NM = 3 * AT * BT;          // MFMAs of a k-step
  constexpr int NH = 2 * (AT + BT);        // half-units of a k-group: (fragment, hi part | lo part)
  // slots (positions behind MFMA m) of a block: group a is converted in [RA0, RA1), group b in [RB0, NM); the slots in
  // front of either cover the latency of the group's fragment reads
  constexpr int RA0 = NM / 6, RA1 = NM / 2, RB0 = NM / 2 + NM / 6;
  Pair sA[2][AT], sB[2][BT];
  f32x4 fa[AT], fb[BT];
  // XR: the k-group's four samples of the extra vector (the same for every lane of a half), its products with the B
  // fragments summed per B row in fp32 (every wave sums, the waves with wa == 0 flush), and its own sum
  f32x4 fe;
  float dsum[BT], esum = 0.f;
#pragma unroll
  for (int j = 0; j < BT; ++j) dsum[j] = 0.f;
  const uint32_t laneX = XR ? lds0 + (uint32_t)it.x_row * 128u : 0u;
  const uint32_t swx = XR ? (uint32_t)((it.x_row >> 1) & 7) : 0u;
  auto reads = [&](int sa, int sb, auto g_) __attribute__((always_inline)) {
    constexpr int GI = decltype(g_)::value;
    read_frags<AT>(fa, laneA + (uint32_t)sa * (uint32_t)kWgSlotBytes + xoff[GI]);
    read_frags<BT>(fb, laneB + (uint32_t)sb * (uint32_t)kWgSlotBytes + xoff[GI]);
    if constexpr (XR) lds_read16<0>(fe, laneX + (uint32_t)sa * (uint32_t)kWgSlotBytes + 16u * ((uint32_t)(2 * GI + h) ^ swx));
  };
  auto landed = [&]() __attribute__((always_inline)) {
    wait_lgkm<0>(); tie_frags<AT>(fa); tie_frags<BT>(fb);
    if constexpr (XR) asm volatile("" : "+v"(fe));
  };
  // half-unit K of the landed k-group -> k-group half HALF of operand set SET (bias sums ride on the A fragments)
  auto half_unit = [&](auto k_, auto half_, auto set_) __attribute__((always_inline)) {
    constexpr int K = decltype(k_)::value, HALF = decltype(half_)::value, SET = decltype(set_)::value;
    constexpr int U = K >> 1, PART = K & 1;
    if constexpr (U < AT) {
      if constexpr (PART == 0) {
        bs[U] += (fa[U][0] + fa[U][1]) + (fa[U][2] + fa[U][3]);
        split4_by_hi<HALF>(sA[SET][U], fa[U], G);
      } else {
        split4_by_lo<HALF>(sA[SET][U], fa[U], G);
      }
    } else {
      if constexpr (PART == 0) {
        if constexpr (XR) {
          float d = dsum[U - AT];
          d = fmaf(fe[0], fb[U - AT][0], d); d = fmaf(fe[1], fb[U - AT][1], d);
          d = fmaf(fe[2], fb[U - AT][2], d); d = fmaf(fe[3], fb[U - AT][3], d);
          dsum[U - AT] = d;
          if constexpr (U == AT) esum += (fe[0] + fe[1]) + (fe[2] + fe[3]);
        }
        split4_hi<HALF>(sB[SET][U - AT], fb[U - AT]);
      } else {
        split4_lo<HALF>(sB[SET][U - AT], fb[U - AT]);
      }
    }
  };
  // One block = the MFMAs of the k-step in operand set SET.  With CONV, k-groups GA and GA+1 of the chunk in ring slots
  // (ra, rb) are read and converted into set 1-SET between them, and, if `go`, operand OP of chunk cs is requested
  // into ring slot ss: pieces 0..3 behind the first MFMAs, 4..7 behind those of the second half.
  auto block = [&](auto set_, auto conv_, auto ga_, int ra, int rb, auto op_, bool go, int cs, int ss) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_)::value, GA = decltype(ga_)::value;
    constexpr bool CONV = decltype(conv_)::value != 0;
    if constexpr (CONV) reads(ra, rb, IC<GA>{});
    static_for<NM>([&](auto m_) __attribute__((always_inline)) {
      constexpr int M = decltype(m_)::value, I = M / (3 * BT), J = (M / 3) % BT, T = M % 3;
      acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(T == 2 ? sA[SET][I].lo : sA[SET][I].hi),
                                                         as_f16x8(T == 1 ? sB[SET][J].lo : sB[SET][J].hi), acc[I][J], 0, 0, 0);
      if constexpr (CONV) {
        static_for<8>([&](auto t_) __attribute__((always_inline)) {
          constexpr int T8 = decltype(t_)::value;
          constexpr int slot = T8 < 4 ? (T8 < NM - 1 ? T8 : NM - 1) : (RA1 + T8 - 4 < NM - 1 ? RA1 + T8 - 4 : NM - 1);
          if constexpr (slot == M) { if (go) stage1(op_, t_, cs, ss); }
        });
        if constexpr (M == RA0) landed();
        static_for<NH>([&](auto k_) __attribute__((always_inline)) {
          if constexpr (RA0 + decltype(k_)::value * (RA1 - RA0) / NH == M) half_unit(k_, IC<0>{}, IC<1 - SET>{});
        });
        if constexpr (M == RA1 - 1) reads(ra, rb, IC<GA + 1>{});
        if constexpr (M == RB0) landed();
        static_for<NH>([&](auto k_) __attribute__((always_inline)) {
          if constexpr (RB0 + decltype(k_)::value * (NM - RB0) / NH == M) half_unit(k_, IC<1>{}, IC<1 - SET>{});
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // the first k-step of the entry is assembled with nothing to hide behind
  reads(0, 1, IC<0>{});
  landed();
  static_for<NH>([&](auto k_) __attribute__((always_inline)) { half_unit(k_, IC<0>{}, IC<0>{}); });
  reads(0, 1, IC<1>{});
  landed();
  static_for<NH>([&](auto k_) __attribute__((always_inline)) { half_unit(k_, IC<1>{}, IC<0>{}); });
  __builtin_amdgcn_sched_barrier(0);
  {
    int sa = 0, sb = 1, pb = 4;   // slots of A(j), B(j), B(j-1)
    for (int j = 0;; ++j) {         // one copy of every converting block; no fragment read in flight at the exit or the back edge
      const bool go = j + 2 < n;
      block(IC<0>{}, IC<1>{}, IC<2>{}, sa, sb, IC<0>{}, go, c0 + j + 2, pb);      // k-step (j, 0); groups (j, 2), (j, 3); A(j+2)
      if (j == n - 1) break;
      // hand-over: chunk j+1 has landed for all and every wave is done reading chunk j
      wait_vm_upto8(go ? nPA : 0);
      __builtin_amdgcn_s_barrier();
      const int nsa = ring_next(sa), nsb = ring_next(sb);
      block(IC<1>{}, IC<1>{}, IC<0>{}, nsa, nsb, IC<1>{}, go, c0 + j + 2, sa);    // k-step (j, 1); groups (j+1, 0), (j+1, 1); B(j+2)
      pb = sb; sa = nsa; sb = nsb;
    }
    block(IC<1>{}, IC<0>{}, IC<0>{}, 0, 0, IC<0>{}, false, 0, 0);                 // the entry's last k-step
  }
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
  if constexpr (XR) {   // the two lane halves hold different samples of the same B row
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      const float v = dsum[j] + __shfl_xor(dsum[j], 32);
      const int col = b_row0 + 32 * j + li;
      if (wa == 0 && h == 0 && col < it.b_rows) gatomic_add(it.dWx + it.col0 + col, v);
    }
    const float e = esum + __shfl_xor(esum, 32);
    if (wave == 0 && lane == 0 && it.dbx != nullptr) gatomic_add(it.dbx, e);
  }
}

#ifdef X3W_STAMPS   // diagnostic build only (scripts/x3w_stamps.py): cycles of every work entry, span of every workgroup
__device__ unsigned long long g_x3w_entry[8192 * 2];
__device__ unsigned long long g_x3w_span[1024 * 2];
#endif
// Persistent workgroup b works through the entries first[b] .. first[b+1]-1 (host: train_impl.inc).
__global__ __launch_bounds__(256, 1) void wgrad_f16x3_kernel(const WgradItem* __restrict__ items,
                                                             const int32_t* __restrict__ first) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // ring of 5 operand-chunk slots
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int e0 = first[blockIdx.x], e1 = first[blockIdx.x + 1];
#ifdef X3W_STAMPS
  if (threadIdx.x == 0) g_x3w_span[2 * blockIdx.x] = wall_clock64();
#endif
  for (int e = e0; e < e1; ++e) {
    const WgradItem it = items[e];
    if (it.c0 >= it.c1) continue;
#ifdef X3W_STAMPS
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    switch (it.shape) {   // (tiles per wave along A) * 16 + (tiles per wave along B), chosen by the host
      case 0x44: wgrad_body<4, 4>(it, smem, wave, lane); break;
      case 0x42: wgrad_body<4, 2>(it, smem, wave, lane); break;
      case 0x41: wgrad_body<4, 1>(it, smem, wave, lane); break;
      case 0x24:
        if (it.x_row >= 0) wgrad_body<2, 4, true>(it, smem, wave, lane);
        else wgrad_body<2, 4>(it, smem, wave, lane);
        break;
      case 0x22: wgrad_body<2, 2>(it, smem, wave, lane); break;
      case 0x21: wgrad_body<2, 1>(it, smem, wave, lane); break;
      case 0x14: wgrad_body<1, 4>(it, smem, wave, lane); break;
      case 0x12: wgrad_body<1, 2>(it, smem, wave, lane); break;
      default:   wgrad_body<1, 1>(it, smem, wave, lane); break;
    }
    __syncthreads();   // every wave is done reading the LDS buffers before the next entry stages into them
#ifdef X3W_STAMPS
    if (threadIdx.x == 0) {
      g_x3w_entry[2 * e] = __builtin_amdgcn_s_memtime() - t0;
      g_x3w_entry[2 * e + 1] = ((unsigned long long)it.a_rows << 48) | ((unsigned long long)it.b_rows << 32) | (unsigned long long)(it.c1 - it.c0) << 12 | blockIdx.x;
    }
#endif
  }
#ifdef X3W_STAMPS
  if (threadIdx.x == 0) g_x3w_span[2 * blockIdx.x + 1] = wall_clock64();
#endif
}

}  // namespace x3

// once per device, with that device current (nerfhip_create): allow the kernel its 160 KiB of dynamic LDS
hipError_t prepare_wgrad_f16x3() {
  return hipFuncSetAttribute((const void*)x3::wgrad_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             x3::kWgSlots * x3::kWgSlotBytes);
}

#ifdef X3W_STAMPS
extern "C" int nerfhip_debug_x3w_stamps(unsigned long long* entry, unsigned long long* span) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(entry, HIP_SYMBOL(x3::g_x3w_entry), sizeof(unsigned long long) * 8192 * 2) != hipSuccess) return 1;
  return hipMemcpyFromSymbol(span, HIP_SYMBOL(x3::g_x3w_span), sizeof(unsigned long long) * 1024 * 2) != hipSuccess;
}
#endif
hipError_t launch_wgrad_f16x3(const WgradItem* d_items, const int32_t* d_first, int ngroups, hipStream_t s) {
  if (ngroups <= 0) return hipSuccess;
  const size_t lds = (size_t)x3::kWgSlots * x3::kWgSlotBytes;  // 160 KiB
  hipLaunchKernelGGL(x3::wgrad_f16x3_kernel, dim3(ngroups), dim3(256), lds, s, d_items, d_first);
  return hipGetLastError();
}

}  // namespace nh
