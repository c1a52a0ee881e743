// mlp_common.hip.h -- device helpers shared by the fused MLP kernels (fp32 MFMA path).
//
// Formulation (all MLP kernels): activations are kept TRANSPOSED, H^T [features][samples], with the
// sample on the MFMA lane and the features in the accumulator registers:
//     H_{l+1}^T (W x 32)  =  W_l (W x K) . H_l^T (K x 32)        per wave, 32 samples per wave
// so the 32x32 accumulator tile of one layer IS the B operand of the next layer's MFMAs with no lane
// movement and no LDS round trip; only the weights (the A operand) stream in (from L2, see below).
//
// v_mfma_f32_32x32x2_f32 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   C/D: lane l, register r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// Register r of an accumulator tile therefore holds feature row R0(r) = (r&3)+8*(r>>2) in lanes 0-31
// and row R0(r)+4 in lanes 32-63: used as B it contracts exactly the k-pair (R0(r), R0(r)+4), and the
// A operand must present W[i][32*kt + R0(r) + 4*(l>>5)].  With r = 4q+j that is k = 32kt + 8q + 4h + j:
// four consecutive k for fixed (q,h), i.e. one 16-byte read per four MFMAs.  Packed weight image ("chunk" =
// 32 k-values x R rows, pack.hip): [ot = i/32][q][lane = 32h + i%32][j]  with k = 8q + 4h + j, so the four
// fragments of one 32x32 output tile are 4 KiB contiguous and each wave-wide 16-byte load reads 1 KiB linearly.
#pragma once
#include "common.h"

namespace nh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NERFHIP_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define NERFHIP_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// acc[ot] += A_chunk(rows 32*ot..) . b   for ot < nt (nt wave-uniform, <= NT); R = rows of the chunk (runtime).
// Plain C++ version (used for the colour head, whose row count is a run-time value).
template <int NT>
__device__ __forceinline__ void mm_chunk(f32x16 (&acc)[NT], const f32x16& b, const float* chunk, int R, int nt,
                                         int li, int h) {
#pragma unroll
  for (int ot = 0; ot < NT; ++ot) {
    if (ot < nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(chunk + (((ot * 4 + q) * 64 + h * 32 + li) * 4));
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[4 * q + 0], acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[4 * q + 1], acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[4 * q + 2], acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[4 * q + 3], acc[ot], 0, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ void mfma16(f32x16& acc, const f32x4 (&a)[4], const f32x16& b) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][0], b[4 * q + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][1], b[4 * q + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][2], b[4 * q + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][3], b[4 * q + 3], acc, 0, 0, 0);
  }
}
// ---------------------------------------------------------------------------------------------------------
// Direct-from-L2 variant (fp32 only): every wave reads its A fragments straight from the packed weight stream
// in global memory (16 B per lane, 1 KiB per wave-instruction, the same image as the LDS chunk) -- no LDS, no
// workgroup barrier, waves never wait for each other.  At the fp32 MFMA rate a wave needs one 1-KiB fragment
// load per 256 cycles: 16 B/clk per CU, ~10 TB/s chip-wide against ~34 TB/s of L2 bandwidth, with the 2.1 MB
// of weights resident in every XCD's 4 MiB L2.
// (The bf16 path cannot do this: 16x the MFMA rate would need 16x the L2 bandwidth -- it keeps the LDS ring.)
//
// The packed stream of a network is ONE flat sequence of 32x32 weight tiles (chunk kt, output tile ot ->
// tile index T = kt*NT + ot, all chunks of the trunk have the same R), so the fragment prefetch runs
// CONTINUOUSLY across layer boundaries: a ring of NB fragment sets lives in the kernel, step T computes tile T
// from set T % NB while tiles T+1 .. T+NB-1 are in flight, and the last NB-1 steps of a layer already fetch the
// first tiles of the next layer (the stream allocation is padded by one chunk so the prefetch past the very
// end stays in bounds).  Every layer part has a tile count that is a multiple of NB, so the ring position is
// the same at every call.  Loads are only ever IN FLIGHT inside a part (pure MFMA + asm code): the compiler
// does not know that the destination registers of an asm load are busy, so at every part boundary the ring is
// landed (land_frags) before compiler-scheduled code runs; scripts/check_inflight.py verifies on the final ISA
// that no instruction touches a register with a load outstanding.
//
// The loads (and the interleaved activation stores of the training kernels, "side" operations) are inline asm
// with hand-counted s_waitcnt vmcnt(N): hipcc (ROCm 7.2) at this register budget sinks ordinary loads to just
// before their use and waits vmcnt(0) there.  vmcnt retires in issue order on gfx9, so VMEM operations the
// kernel does NOT count (the compiler's own loads/stores, side operations of the previous call) can only make
// a counted wait stricter, never too weak; every operation that IS counted is an asm volatile statement and
// therefore stays in program order.
//
// HAZARD NOTE: "VALU writes SGPR -> VMEM reads that SGPR" needs 5 wait states on gfx9 (CDNA ISA, manually
// inserted wait states).  hipcc inserts them for its own instructions but does not look inside inline asm, and
// it does reload spilled SGPR bases with v_readlane right in front of an asm statement.  So every asm statement
// here that reads an SGPR address first copies it with s_mov_b64 into a pair of its own: the SALU read of a
// VALU-written SGPR is interlocked, and so is the VMEM read of the SALU-written copy.  (The first version put
// s_nop 4 there instead: ~16 cycles more per statement than one SALU instruction -- with 128 stores per layer and
// wave that was 2.4 % of the fp32 training forward and 5 % of the split-f16 one.)
template <int T>   // tile T of the part that starts at base: 4 KiB at base + T*4 KiB, fragment q at +q KiB
__device__ __forceinline__ void gload_tile(f32x4 (&a)[4], const float* base, uint32_t voff) {
  uint64_t sb;
  asm volatile(
      "s_mov_b64 %4, %6\n\t"
      "global_load_dwordx4 %0, %5, %4\n\t"
      "global_load_dwordx4 %1, %5, %4 offset:1024\n\t"
      "global_load_dwordx4 %2, %5, %4 offset:2048\n\t"
      "global_load_dwordx4 %3, %5, %4 offset:3072"
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&s"(sb)
      : "v"(voff), "s"(base + T * 1024));
}
__device__ __forceinline__ void gload_frag(f32x4& a, const float* sbase, uint32_t voff) {
  uint64_t sb;
  asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1" : "=&v"(a), "=&s"(sb) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void gload4u(uint32_t& d, const uint32_t* sbase, uint32_t voff) {
  uint64_t sb;
  asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dword %0, %2, %1" : "=&v"(d), "=&s"(sb) : "v"(voff), "s"(sbase));
}
template <int OFF>   // immediate byte offset, < 4096
__device__ __forceinline__ void gstore4(float* sbase, uint32_t voff, float v) {
  static_assert(OFF >= 0 && OFF < 4096, "global immediate offset");
  // nt: the stored activations / gradients are read once, much later, by another kernel; non-temporal keeps them from
  // pushing the packed weight stream (which every wave re-reads from L2) out of the XCD's L2
  uint64_t sb;
  asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dword %1, %2, %0 offset:%4 nt" : "=&s"(sb) : "v"(voff), "v"(v), "s"(sbase), "n"(OFF));
}
constexpr int kNB = 4;   // fragment ring: 1 computing + 3 tiles (768 MFMA-cycles) in flight

// side operations: ops(T) = number of VMEM instructions issue<T>() emits (exact, compile time)
// after_wait<T>() runs right behind step T's counted wait (used to pin registers of side LOADS until they landed)
struct NoSide {
  static constexpr int ops(int) { return 0; }
  template <int T> __device__ __forceinline__ void issue() const {}
  template <int T> __device__ __forceinline__ void after_wait() const {}
};
// Stores into the tile-major activation buffers (common.h): `dst` is the wave's 32-sample block of a buffer
// (wave-uniform -> SGPR), register r of tile t goes to row 32t + R0(r) (+4 in the upper lane half), i.e. to
//     dst + t*4096 B + R0(r)*128 B (immediate)  +  voff,   voff = tile_voff(lane) = (128 h + li) * 4
// -- one SGPR base per 32 rows and no per-row address arithmetic.
__device__ __forceinline__ uint32_t tile_voff(int lane) { return (uint32_t)((128 * (lane >> 5) + (lane & 31)) * 4); }
__device__ __forceinline__ float* tile_block(float* buf, int tix, int rows) { return buf + (int64_t)tix * rows * 32; }
template <int T_, int R_>
__device__ __forceinline__ void store_reg(float* dst, uint32_t voff, float v) {
  gstore4<((R_ & 3) + 8 * (R_ >> 2)) * 128>(dst + T_ * 1024, voff, v);
}
// spreads the stores of the NSRC*16 registers of src evenly over TOTAL steps
template <int NSRC, int TOTAL>
struct StoreSide {
  static constexpr int N = NSRC * 16;
  static constexpr int PER = (N + TOTAL - 1) / TOTAL;
  const f32x16* src;
  float* dst;
  uint32_t voff;
  static constexpr int ops(int T) { return N - T * PER <= 0 ? 0 : (N - T * PER < PER ? N - T * PER : PER); }
  template <int IDX> __device__ __forceinline__ void one() const {
    if constexpr (IDX < N) store_reg<IDX / 16, IDX % 16>(dst, voff, src[IDX / 16][IDX % 16]);
  }
  template <int T> __device__ __forceinline__ void after_wait() const {}
  template <int T> __device__ __forceinline__ void issue() const {
    static_assert(PER == 1 || PER == 2 || PER == 4 || PER == 8 || PER == 16, "stores per step must be a power of two");
    one<T * PER + 0>(); if constexpr (PER > 1) one<T * PER + 1>();
    if constexpr (PER > 2) { one<T * PER + 2>(); one<T * PER + 3>(); }
    if constexpr (PER > 4) { one<T * PER + 4>(); one<T * PER + 5>(); one<T * PER + 6>(); one<T * PER + 7>(); }
    if constexpr (PER > 8) {
      one<T * PER + 8>(); one<T * PER + 9>(); one<T * PER + 10>(); one<T * PER + 11>();
      one<T * PER + 12>(); one<T * PER + 13>(); one<T * PER + 14>(); one<T * PER + 15>();
    }
  }
};
// all rows of the first nt (<= NT) tiles of src at once
template <int NT, int T_ = 0>
__device__ __forceinline__ void store_tiles(float* dst, const f32x16* src, int nt, uint32_t voff) {
  if constexpr (T_ < NT) {
    if (T_ < nt) {
      store_reg<T_, 0>(dst, voff, src[T_][0]);   store_reg<T_, 1>(dst, voff, src[T_][1]);
      store_reg<T_, 2>(dst, voff, src[T_][2]);   store_reg<T_, 3>(dst, voff, src[T_][3]);
      store_reg<T_, 4>(dst, voff, src[T_][4]);   store_reg<T_, 5>(dst, voff, src[T_][5]);
      store_reg<T_, 6>(dst, voff, src[T_][6]);   store_reg<T_, 7>(dst, voff, src[T_][7]);
      store_reg<T_, 8>(dst, voff, src[T_][8]);   store_reg<T_, 9>(dst, voff, src[T_][9]);
      store_reg<T_, 10>(dst, voff, src[T_][10]); store_reg<T_, 11>(dst, voff, src[T_][11]);
      store_reg<T_, 12>(dst, voff, src[T_][12]); store_reg<T_, 13>(dst, voff, src[T_][13]);
      store_reg<T_, 14>(dst, voff, src[T_][14]); store_reg<T_, 15>(dst, voff, src[T_][15]);
    }
    store_tiles<NT, T_ + 1>(dst, src, nt, voff);
  }
}
template <class Side>
constexpr int side_ops_sum(int first, int last) {
  int n = 0;
  for (int k = first < 0 ? 0 : first; k <= last; ++k) n += Side::ops(k);
  return n;
}
template <int CNT>
__device__ __forceinline__ void wait_frag(f32x4 (&a)[4]) {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(CNT));
}
// every ring register has landed after this (it also orders the ring against whatever the compiler does next)
__device__ __forceinline__ void land_frags(f32x4 (&f)[kNB][4]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[0][3]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]),
                 "+v"(f[1][3]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[2][2]), "+v"(f[2][3]), "+v"(f[3][0]), "+v"(f[3][1]),
                 "+v"(f[3][2]), "+v"(f[3][3]));
  __builtin_amdgcn_sched_barrier(0);   // ... and nothing of what follows may be hoisted into the in-flight window
}
// SPR: the four loads of tile T+PD go out one per four MFMAs of step T instead of all in front of them.  Measured
// (round 1): the backward kernel gains 3 % (7.88 -> 7.64 ms per step), the forward kernels LOSE 2-3 %, so only the
// backward kernel asks for it.
template <int NT, int R, int KT, class Side, int T, bool SPR>
struct LStep {
  static __device__ __forceinline__ void run(f32x16 (&acc)[NT], const f32x16* bt, f32x4 (&f)[kNB][4], const float* base,
                                             uint32_t voff, const Side& side) {
    constexpr int PD = kNB - 1;
    side.template issue<T>();
    if constexpr (!SPR) {
      gload_tile<T + PD>(f[(T + PD) % kNB], base, voff);   // past the end of this part = first tiles of the next
      // in flight behind tile T's loads: tiles T+1..T+PD and the side operations of steps T-PD+1..T
      wait_frag<4 * PD + side_ops_sum<Side>(T - PD + 1, T)>(f[T % kNB]);
      side.template after_wait<T>();
      mfma16(acc[T % NT], f[T % kNB], bt[T / NT]);
    } else {
      // in flight behind tile T's loads: tiles T+1..T+PD-1 and the side operations of steps T-PD+1..T
      wait_frag<4 * (PD - 1) + side_ops_sum<Side>(T - PD + 1, T)>(f[T % kNB]);
      side.template after_wait<T>();
      f32x16& c = acc[T % NT];
      const f32x4 (&a)[4] = f[T % kNB];
      f32x4 (&n)[4] = f[(T + PD) % kNB];
      const f32x16& b = bt[T / NT];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gload_frag(n[q], base + (T + PD) * 1024 + q * 256, voff);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][0], b[4 * q + 0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][1], b[4 * q + 1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][2], b[4 * q + 2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][3], b[4 * q + 3], c, 0, 0, 0);
      }
    }
    if constexpr (T + 1 < KT * NT) LStep<NT, R, KT, Side, T + 1, SPR>::run(acc, bt, f, base, voff, side);
    // Part boundary: the tiles fetched ahead for the next part were issued 1..3 steps (1024 cycles each) ago and
    // have all but certainly landed while this step's 16 MFMAs drain, so this wait is nearly free -- and from
    // here on the ring holds ordinary values the compiler may copy or spill as it likes.
    else land_frags(f);
  }
};
__device__ __forceinline__ uint32_t frag_voff(int lane) { return (uint32_t)(lane * 16); }
// first call on a stream: fetch tiles 0..NB-2; must be followed directly by mm_layer_g
__device__ __forceinline__ void prime_frags(f32x4 (&f)[kNB][4], const float* base, uint32_t voff) {
  __builtin_amdgcn_sched_barrier(0);   // nothing of the prologue may sink below the first in-flight load
  gload_tile<0>(f[0], base, voff);
  gload_tile<1>(f[1], base, voff);
  gload_tile<2>(f[2], base, voff);
}
// acc[ot] += sum_kt W[:, chunk kt] . bt[kt];  base -> first chunk of this part (global memory, packed image);
// the ring must already hold / be fetching tiles 0..NB-2 of this part (prime_frags or the previous part).
template <int NT, int R, int KT, bool SPR = false, class Side>
__device__ __forceinline__ void mm_layer_g(f32x16 (&acc)[NT], const f32x16* bt, f32x4 (&f)[kNB][4], const float* base,
                                           uint32_t voff, const Side& side) {
  static_assert((KT * NT) % kNB == 0, "a layer part must keep the ring position");
  LStep<NT, R, KT, Side, 0, SPR>::run(acc, bt, f, base, voff, side);
}

// feature row held by register r of tile t in lane-half h
__device__ __forceinline__ constexpr int tile_row(int t, int r, int h) {
  return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// byte offset of lane (h, sample m) inside one [4 words][Mp] plane pair of the ReLU mask buffer
__device__ __forceinline__ uint32_t mask_voff(int h, int64_t Mp, int64_t m) { return (uint32_t)((4 * h * Mp + m) * 4); }
__device__ __forceinline__ constexpr int tile_row_u(int t, int r) { return 32 * t + (r & 3) + 8 * (r >> 2); }

// ReLU sign bits of NT accumulator tiles, 2 tiles per 32-bit word
template <int NT>
__device__ __forceinline__ void relu_mask_words(const f32x16 (&a)[NT], uint32_t (&w)[4], int nt) {
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = 0u;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {   // "> 0" on the integer bits (same order as the floats'): clamp to 0/1, shift-or
        const float x = a[t][r];        // (a temporary: bit_cast of a vector ELEMENT expression reads element 0 with this clang)
        w[t >> 1] |= (uint32_t)min(max(__builtin_bit_cast(int, x), 0), 1) << (16 * (t & 1) + r);
      }
    }
  }
}
__device__ __forceinline__ uint32_t* mask_ptr(uint32_t* mask, int slot, int h, int w, int64_t Mp, int64_t m) {
  return mask + ((int64_t)(slot * 8 + h * 4 + w)) * Mp + m;
}

// acc[t][4q..4q+3] = v[32t + 8q + 4h .. +3]   (v has >= 32*nt floats, 16-byte aligned rows of 4)
template <int NT>
__device__ __forceinline__ void load_rows(f32x16 (&acc)[NT], const float* v, int nt, int nvalid, int h) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 32 * t + 8 * q + 4 * h;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (t < nt && row + 3 < nvalid) x = *reinterpret_cast<const f32x4*>(v + row);
      else if (t < nt && row < nvalid) {
        x[0] = v[row];
        if (row + 1 < nvalid) x[1] = v[row + 1];
        if (row + 2 < nvalid) x[2] = v[row + 2];
      }
      acc[t][4 * q + 0] = x[0]; acc[t][4 * q + 1] = x[1]; acc[t][4 * q + 2] = x[2]; acc[t][4 * q + 3] = x[3];
    }
  }
}

// One value of positionCode(x, L) (nerfCoreBlock.scala:127-136): index e of [x, sin(x), cos(x), sin(2x) ...]
// sin(a) or cos(a) to ~1.5 ulp of a value near 1 (absolute error <= 1e-7, exact at a = 0) for |a| <= kTrigMax:
// Cody-Waite reduction by pi/2 in three fused steps (the first is exact: a - n fl(pi/2) is representable), cephes'
// single-precision kernels on [-pi/4, pi/4], cos(a) = sin(a + pi/2).  About 20 instructions; libm's sinf/cosf carry
// a Payne-Hanek reduction for huge arguments that the compiler evaluates for EVERY value (selects, not branches):
// 235 instructions per value, 20 000 cycles per 128-sample tile for the 32 values a lane encodes.
constexpr float kTrigMax = 8192.0f;
__device__ __forceinline__ float sin_or_cos_reduced(float a, int want_cos) {   // want_cos: 0 | 1
  const float n = __builtin_rintf(a * 0.636619772f);
  float r = fmaf(n, -1.5707963705062866f, a);
  r = fmaf(n, 4.371138828673793e-08f, r);
  r = fmaf(n, 1.7151245100058819e-15f, r);
  const int q = (int)n + want_cos;
  const float z = r * r;
  const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  const float sn = fmaf(ps * z, r, r);
  const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  const float cs = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
  const float v = (q & 1) ? cs : sn;
  return (q & 2) ? -v : v;
}

// Element e of positionCode(p, L) (nerfCoreBlock.scala:127-136): [p, sin(p 2^0), cos(p 2^0), ..., cos(p 2^(L-1))], each
// block 3 wide.  The fused kernels give lane half h the elements e = e0 + 4 h with e0 known at compile time (after
// unrolling), so what an element is -- coordinate, frequency, sin or cos -- is a choice between two constants and both
// lane halves run ONE instruction stream (a divergent `if (h)` around two specialised streams costs both).
// RED: every argument of the wave's samples is within kTrigMax (trig_reduced), else libm's functions.
struct CodeInfo { int coord; float scale; int want_cos; bool raw; };
__host__ __device__ constexpr CodeInfo code_info(int e) {
  if (e < 3) return CodeInfo{e, 1.f, 0, true};
  const int t = e - 3, f = t / 6, rem = t - 6 * f;
  return CodeInfo{rem >= 3 ? rem - 3 : rem, (float)(1 << f), rem >= 3 ? 1 : 0, false};   // input.mul(1 << i)
}
template <bool RED>
__device__ __forceinline__ float code_value_h(float px, float py, float pz, int e0, int h, int E) {
  const CodeInfo i0 = code_info(e0), i1 = code_info(e0 + 4);
  const float x0 = i0.coord == 0 ? px : (i0.coord == 1 ? py : pz);
  const float x1 = i1.coord == 0 ? px : (i1.coord == 1 ? py : pz);
  const float x = h ? x1 : x0;
  const float a = x * (h ? i1.scale : i0.scale);
  const int wc = h ? i1.want_cos : i0.want_cos;
  float v;
  if constexpr (RED) {
    v = sin_or_cos_reduced(a, wc);
  } else {
    float sn, cs;
    sincosf(a, &sn, &cs);
    v = wc ? cs : sn;
  }
  if (i0.raw) v = h ? v : x0;            // e0 < 3: the coordinate itself (e0 + 4 never is)
  return e0 + 4 * h < E ? v : 0.f;       // zero padding up to 64 rows
}
// wave-uniform: all encoding arguments p * 2^(L-1) of the wave's samples are within the reduced routine's range (E = 3 + 6 L)
__device__ __forceinline__ bool trig_reduced(float px, float py, float pz, int E) {
  const float top = (float)(1 << ((E - 3) / 6 > 0 ? (E - 3) / 6 - 1 : 0));
  const bool ok = fmaxf(fmaxf(fabsf(px), fabsf(py)), fabsf(pz)) * top <= kTrigMax;   // false for NaN too
  return __builtin_amdgcn_ballot_w64(!ok) == 0;
}

}  // namespace nh
