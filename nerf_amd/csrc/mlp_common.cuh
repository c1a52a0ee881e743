// mlp_common.cuh -- device helpers shared by the fused MLP kernels (fp32 MFMA path).
//
// Formulation (all MLP kernels): activations are kept TRANSPOSED, H^T [features][samples], with the
// sample on the MFMA lane and the features in the accumulator registers:
//     H_{l+1}^T (W x 32)  =  W_l (W x K) . H_l^T (K x 32)        per wave, 32 samples per wave
// so the 32x32 accumulator tile of one layer IS the B operand of the next layer's MFMAs with no lane
// movement and no LDS round trip; only the weights (the A operand) stream through LDS.
//
// v_mfma_f32_32x32x2_f32 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   C/D: lane l, register r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// Register r of an accumulator tile therefore holds feature row R0(r) = (r&3)+8*(r>>2) in lanes 0-31
// and row R0(r)+4 in lanes 32-63: used as B it contracts exactly the k-pair (R0(r), R0(r)+4), and the
// A operand must present W[i][32*kt + R0(r) + 4*(l>>5)].  With r = 4q+j that is k = 32kt + 8q + 4h + j:
// four consecutive k for fixed (q,h), i.e. one 16-byte LDS read per four MFMAs from a chunk image laid
// out [g = k/4 (8)][i (R)][k%4 (4)]  (g = 2q + h).
#pragma once
#include "common.h"

namespace nh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NERFHIP_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define NERFHIP_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// Asynchronous global -> LDS copy of one chunk (nfloats*4 bytes, a multiple of 4096): every wave moves
// 1 KiB per instruction (16 B per lane, LDS destination = wave-uniform base + lane*16).
__device__ __forceinline__ void stage_chunk(const float* g, float* lds, uint32_t nfloats, int wave, int lane) {
  const uint32_t bytes = nfloats * 4u;
  for (uint32_t off = (uint32_t)wave * 1024u; off < bytes; off += 4096u) {
    __builtin_amdgcn_global_load_lds(NERFHIP_GPTR((const char*)g + off + lane * 16),
                                     NERFHIP_LPTR((char*)lds + off), 16, 0, 0);
  }
}

// Double-buffered weight-chunk pipeline.  acquire() makes chunk c visible to the whole workgroup (own DMA
// landed + barrier: everyone is also done reading chunk c-1); prefetch() starts the DMA of chunk c+1 into
// the other buffer and is called a few MFMAs INTO chunk c, so that the matrix pipe is already busy while the
// DMA instructions issue.  Chunk offsets are arithmetic (no table loads on the critical path): the first
// n_main chunks have cs_main floats each, the rest (the colour head of the forward stream) cs_tail.
// All 256 threads must call both.
struct ChunkPipe {
  const float* wpack;
  float* buf0;
  float* buf1;
  uint32_t cs_main, cs_tail;
  int n_main, n, c, wave, lane;
  __device__ __forceinline__ const float* src(int i) const {
    return wpack + (i < n_main ? (uint32_t)i * cs_main : (uint32_t)n_main * cs_main + (uint32_t)(i - n_main) * cs_tail);
  }
  __device__ __forceinline__ void prime(int first) {
    c = first;
    if (c < n) stage_chunk(src(c), (c & 1) ? buf1 : buf0, c < n_main ? cs_main : cs_tail, wave, lane);
  }
  __device__ __forceinline__ const float* acquire() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return (c & 1) ? buf1 : buf0;
  }
  __device__ __forceinline__ void prefetch() {
    if (c + 1 < n) stage_chunk(src(c + 1), (c & 1) ? buf0 : buf1, (c + 1) < n_main ? cs_main : cs_tail, wave, lane);
    ++c;
  }
  __device__ __forceinline__ const float* next() {  // acquire + immediate prefetch
    const float* cur = acquire();
    prefetch();
    return cur;
  }
};

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
        const f32x4 a = *reinterpret_cast<const f32x4*>(chunk + ((2 * q + h) * R + 32 * ot + li) * 4);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[4 * q + 0], acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[4 * q + 1], acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[4 * q + 2], acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[4 * q + 3], acc[ot], 0, 0, 0);
      }
    }
  }
}

// Software-pipelined version for compile-time R (the trunk): hipcc (ROCm 7.2), at the register budget of
// these kernels, sinks every ds_read_b128 to just before its 4 MFMAs and waits lgkmcnt(0) there (~25 % of the
// matrix pipe idle).  Here the four A-fragment reads of tile ot+1 are issued (inline asm, immediate offsets
// from one base VGPR) BEFORE the 16 MFMAs of tile ot, and each tile waits with a counted lgkmcnt(4) whose
// "+v" operands tie the fragments to the wait (cdna_hip_programming.md 5.7 item 1, form ii).  Counts stay
// conservative if the compiler has LDS/SMEM operations of its own in flight (LDS returns in order).
template <int IMM>
__device__ __forceinline__ void ds_read16(f32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int R, int OT>
__device__ __forceinline__ void read_tile(f32x4 (&a)[4], uint32_t base) {
  ds_read16<(0 * 2 * R + 32 * OT) * 16>(a[0], base);
  ds_read16<(1 * 2 * R + 32 * OT) * 16>(a[1], base);
  ds_read16<(2 * 2 * R + 32 * OT) * 16>(a[2], base);
  ds_read16<(3 * 2 * R + 32 * OT) * 16>(a[3], base);
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
// Two output tiles per step with their MFMAs interleaved (two independent accumulator chains), fragments of
// the next pair prefetched: [reads pair p+1] [wait pair p] [32 MFMAs alternating acc[2p], acc[2p+1]].
__device__ __forceinline__ void mfma16x2(f32x16& c0, f32x16& c1, const f32x4 (&a0)[4], const f32x4 (&a1)[4],
                                         const f32x16& b) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q][j], b[4 * q + j], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q][j], b[4 * q + j], c1, 0, 0, 0);
    }
  }
}
template <int NT, int R, int P>   // P = pair index, NT even
struct MMPair {
  static __device__ __forceinline__ void run(f32x16 (&acc)[NT], f32x4 (&c0)[4], f32x4 (&c1)[4], f32x4 (&n0)[4],
                                             f32x4 (&n1)[4], const f32x16& b, uint32_t base, ChunkPipe& pipe) {
    if constexpr (2 * P + 2 < NT) {
      read_tile<R, 2 * P + 2>(n0, base);
      read_tile<R, 2 * P + 3>(n1, base);
      asm volatile("s_waitcnt lgkmcnt(8)"
                   : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]));
    }
    mfma16x2(acc[2 * P], acc[2 * P + 1], c0, c1, b);
    if constexpr (P == 0) pipe.prefetch();   // DMA of the next chunk issues behind MFMAs already in the pipe
    if constexpr (2 * P + 2 < NT) MMPair<NT, R, P + 1>::run(acc, n0, n1, c0, c1, b, base, pipe);
  }
};
template <int NT, int R>
__device__ __forceinline__ void mm_chunk_t(f32x16 (&acc)[NT], const f32x16& b, ChunkPipe& pipe, int li, int h) {
  const float* chunk = pipe.acquire();
  const uint32_t base =
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)chunk + (uint32_t)((h * R + li) * 16);
  if constexpr (NT == 1) {
    f32x4 a0[4];
    read_tile<R, 0>(a0, base);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]));
    mfma16(acc[0], a0, b);
    pipe.prefetch();
  } else {
    f32x4 a0[4], a1[4], n0[4], n1[4];
    read_tile<R, 0>(a0, base);
    read_tile<R, 1>(a1, base);
    MMPair<NT, R, 0>::run(acc, a0, a1, n0, n1, b, base, pipe);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Direct-from-L2 variant (fp32 only): every wave reads its A fragments straight from the packed weight stream
// in global memory (16 B per lane, 1 KiB per wave-instruction, the same image as the LDS chunk) -- no LDS, no
// workgroup barrier, waves never wait for each other.  At the fp32 MFMA rate a wave needs one 1-KiB fragment
// load per 256 cycles: 16 B/clk per CU, ~10 TB/s chip-wide against ~34 TB/s of L2 bandwidth, with the 2.1 MB
// of weights resident in every XCD's 4 MiB L2.  Fragments are fetched two tiles (2048 MFMA-cycles) ahead.
// (The bf16 path cannot do this: 16x the MFMA rate would need 16x the L2 bandwidth -- it keeps the LDS ring.)
// One LAYER-part = KT chunks x NT tiles, walked as a flat tile stream T = kt*NT + ot with three rotating
// fragment sets (tile T computing, T+1 and T+2 in flight).  bt[kt] is the B operand of chunk kt.
// The loads are inline asm (SGPR base + 32-bit lane offset) with hand-counted vmcnt waits, because hipcc
// (ROCm 7.2) at this register budget sinks ordinary loads to just before their use and waits vmcnt(0) there.
// vmcnt retires in issue order, so operations of the compiler's own (bias loads, activation stores) that
// are in flight can only make these counted waits stricter, never too weak.
__device__ __forceinline__ void gload16(f32x4& d, const float* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase));
}
template <int R, int NT, int T>
__device__ __forceinline__ void gload_tile(f32x4 (&a)[4], const float* base, uint32_t voff) {
  constexpr int kt = T / NT, ot = T % NT;
  gload16(a[0], base + kt * 32 * R + (0 * 2 * R + 32 * ot) * 4, voff);
  gload16(a[1], base + kt * 32 * R + (1 * 2 * R + 32 * ot) * 4, voff);
  gload16(a[2], base + kt * 32 * R + (2 * 2 * R + 32 * ot) * 4, voff);
  gload16(a[3], base + kt * 32 * R + (3 * 2 * R + 32 * ot) * 4, voff);
}
template <int NT, int R, int KT, int T>
struct LStep {
  static __device__ __forceinline__ void run(f32x16 (&acc)[NT], const f32x16 (&bt)[KT], f32x4 (&f0)[4], f32x4 (&f1)[4],
                                             f32x4 (&f2)[4], const float* base, uint32_t voff) {
    constexpr int total = KT * NT;
    if constexpr (T + 2 < total) {
      gload_tile<R, NT, T + 2>(f2, base, voff);
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]));
    } else if constexpr (T + 1 < total) {
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]));
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]));
    }
    mfma16(acc[T % NT], f0, bt[T / NT]);
    if constexpr (T + 1 < total) LStep<NT, R, KT, T + 1>::run(acc, bt, f1, f2, f0, base, voff);
  }
};
// acc[ot] += sum_kt W[:, chunk kt] . bt[kt];  base -> first chunk (global memory, packed image).
template <int NT, int R, int KT>
__device__ __forceinline__ void mm_layer_g(f32x16 (&acc)[NT], const f32x16 (&bt)[KT], const float* base, int li, int h) {
  const uint32_t voff = (uint32_t)((h * R + li) * 16);
  f32x4 f0[4], f1[4], f2[4];
  gload_tile<R, NT, 0>(f0, base, voff);
  if constexpr (KT * NT > 1) gload_tile<R, NT, 1>(f1, base, voff);
  LStep<NT, R, KT, 0>::run(acc, bt, f0, f1, f2, base, voff);
}

// feature row held by register r of tile t in lane-half h
__device__ __forceinline__ constexpr int tile_row(int t, int r, int h) {
  return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
}

// Feature-major activation store addressing: element (row, m) of a [rows][Mp] fp32 buffer where
// row = row_u + 4h (row_u wave-uniform).  Written as (uniform 64-bit base) + (32-bit per-lane byte offset)
// so the compiler emits SGPR-base global loads/stores instead of one 64-bit VGPR address per row.
__device__ __forceinline__ uint32_t lane_off(int h, int64_t Mp, int64_t m) { return (uint32_t)((4 * h * Mp + m) * 4); }
__device__ __forceinline__ float* row_ptr(float* base, int64_t row_u, int64_t Mp, uint32_t voff) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base + row_u * Mp) + voff);
}
__device__ __forceinline__ const float* row_ptr(const float* base, int64_t row_u, int64_t Mp, uint32_t voff) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + row_u * Mp) + voff);
}
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
      for (int r = 0; r < 16; ++r) w[t >> 1] |= (a[t][r] > 0.f ? 1u : 0u) << (16 * (t & 1) + r);
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
__device__ __forceinline__ float code_value(float px, float py, float pz, int e, int E) {
  if (e >= E) return 0.f;
  if (e < 3) return e == 0 ? px : (e == 1 ? py : pz);
  const int t = e - 3;
  const int f = t / 6;
  const int rem = t - 6 * f;
  const int c = rem >= 3 ? rem - 3 : rem;
  const float x = c == 0 ? px : (c == 1 ? py : pz);
  const float a = x * (float)(1 << f);   // input.mul(1 << i)
  float s, co;
  sincosf(a, &s, &co);
  return rem >= 3 ? co : s;
}

}  // namespace nh
