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

// Double-buffered weight-chunk pipeline: next() makes chunk c visible to the whole workgroup and starts
// the DMA of chunk c+1 into the other buffer.  Must be called by all 256 threads.
struct ChunkPipe {
  const float* wpack;
  const uint32_t* off;
  float* buf0;
  float* buf1;
  int n, c, wave, lane;
  __device__ __forceinline__ void prime() {
    c = 0;
    if (n > 0) stage_chunk(wpack + off[0], buf0, off[1] - off[0], wave, lane);
  }
  __device__ __forceinline__ const float* next() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own DMA pieces of chunk c have landed
    __syncthreads();                                   // everyone's have; everyone is done with chunk c-1
    const float* cur = (c & 1) ? buf1 : buf0;
    if (c + 1 < n) stage_chunk(wpack + off[c + 1], (c & 1) ? buf0 : buf1, off[c + 2] - off[c + 1], wave, lane);
    ++c;
    return cur;
  }
};

// acc[ot] += A_chunk(rows 32*ot..) . b   for ot < nt (nt wave-uniform, <= NT)
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
