// wgrad_bf16.hip -- weight-gradient GEMM of the MLP on bf16 MFMA (gfx950), fp32 accumulation and output.
//
//   dW[i][j] = sum_m  A[i][m] * B[j][m]          A = dY (gradient rows), B = X (input feature rows)
// Both operands come from the bf16 training store (common.h: BfStore) in MFMA-operand layout of the FORWARD
// products, i.e. sample on the lane, 8 features per lane.  This product contracts over the SAMPLE axis, so every
// fragment is transposed on its way out of LDS with ds_read_b64_tr_b16 (cdna_hip_programming.md T10): within a
// 16-lane group, lane 4q+p supplies the address of (sample m0+q, features 4p'..4p'+3) -- 8 contiguous bytes of
// the stored image -- and lane c receives feature c of samples m0..m0+3; two such reads give the 8 consecutive
// samples (k = 8h .. 8h+7) of one v_mfma_f32_32x32x16_bf16 operand row.
// Same work decomposition as the fp32 kernel (wgrad.hip): persistent workgroups, one entry = a slice of the sample
// axis of one product, a <=256x256 output owned by the workgroup in accumulators, the 4 waves arranged 2x2 / 1x4 /
// 4x1, fp32 atomics into the canonical gradient buffer.  At bf16 MFMA speed the kernel is HBM-bound (one 32-sample
// chunk of a 256x256 product = 32 KiB of operands for 32 MFMAs per wave), so the staging ring is 4 deep with
// counted vmcnt waits, and the staged image is swizzled (wb_slot) so the transposed reads are bank-conflict free.
// Bias gradients (row sums of A) are v_dot2_f32_bf16 sums over the A fragments a wave holds anyway.
#include "bf16_common.hip.h"
#include "launchers.h"

namespace nh {

constexpr int kWbStages = 5;                  // 5 x 32 KiB = the CU's whole LDS
constexpr int kWbAhead = kWbStages - 2;       // whole chunks in flight behind the one being waited for
constexpr int kWbStageBytes = 32 * 1024;      // A tiles (<= 8 x 2 KiB) | B tiles (<= 8 x 2 KiB)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;

// LDS image of one staged fragment (64 lanes x 16 B, lane = 32*hp + sample): the 16-byte slot of (hp, sample m) is
//   32*hp + (m & 16) + ((m + 8*s + 4*hp) & 15),   s = the fragment's k-step
// so that the 32 eight-byte pieces one half-wave fetches in a transposed read (2 k-steps x 2 lane halves x 4
// samples x 2 element halves) fall into 32 different bank pairs; the linear image has them 4 deep on 8 pairs.
__device__ __forceinline__ int wb_slot(int s, int hp, int m) { return 32 * hp + (m & 16) + ((m + 8 * s + 4 * hp) & 15); }

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int N>
struct WbOps { bf16x8 v[N]; };
template <int N>
struct WbRaw { u32x2 lo[N], hi[N]; };   // destinations of transposed reads in flight

// PW: staging pieces (1 KiB each) a wave issues per chunk -- 4 or 8; a wave with fewer real pieces re-issues its
// last one (same source, same destination), so that every wait count is a compile-time constant.
template <int AT, int BT, int PW>
__device__ __forceinline__ void wgrad_bf16_body(const WgradBfItem& it, char* smem, int wave, int lane) {
  const int li = lane & 31, h = lane >> 5;
  const int wa = wave / it.wb_n, wb = wave - wa * it.wb_n;
  const int a_tile0 = AT * wa, b_tile0 = BT * wb;          // first 32-row tile of this wave
  const int a_tiles_all = (it.a_rows + 31) >> 5, b_tiles_all = (it.b_rows + 31) >> 5;
  const int c0 = it.c0, c1 = it.c1, cl = c1 - 1;

  f32x16 acc[AT][BT];
  float bsum[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < BT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // the waves that share an A tile split its bias sums: tile i belongs to the wave with wb == i mod wb_n
  bool do_bias[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) do_bias[i] = it.db != nullptr && (i & (it.wb_n - 1)) == wb;

  // staging: one stage = the chunk's A tiles (2 KiB each, contiguous in the store) then its B tiles, piece idx of
  // the stage = wave + 4p.  Per piece: a running source pointer, its stride per chunk and its offset in the stage.
  const int pieces_total = 2 * (a_tiles_all + b_tiles_all);
  const int idx_last = wave + 4 * ((pieces_total - 1 - wave) >> 2);    // pieces_total >= 4: every wave has one
  const char* pptr[PW];
  uint32_t pstr[PW], pdst[PW];
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int idx = min(wave + 4 * p, idx_last);
    const bool isb = idx >= 2 * a_tiles_all;
    const int k = isb ? idx - 2 * a_tiles_all : idx;
    pstr[p] = (uint32_t)(isb ? it.b_stride : it.a_stride);
    pptr[p] = reinterpret_cast<const char*>(isb ? it.B : it.A) + (int64_t)c0 * pstr[p] + k * 1024;
    pdst[p] = (isb ? 16384u : 0u) + (uint32_t)k * 1024u;
  }
  // LDS slot `lane` of a piece takes the source lane that wb_slot() maps there; the k-step of every piece of this
  // wave is wave & 1
  const uint32_t src_lane = (uint32_t)(32 * h + (li & 16) + (((li & 15) - 8 * (wave & 1) - 4 * h) & 15)) * 16u;
  // piece p of chunk min(r, cl) into ring stage (r - c0) & 3; called with consecutive r for every piece
  auto issue = [&](int p, int r) {
    char* d = smem + (uint32_t)(r - c0) % (uint32_t)kWbStages * kWbStageBytes + pdst[p];
    __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(pptr[p] + src_lane), NERFHIP_LPTR(d), 16, 0, 0);
    pptr[p] += r < cl ? pstr[p] : 0u;
  };
  // transposed fragment reads: group g = lane>>4 -> (sample half hg = g>>1, k-step s = g&1); lane 4q+p of the
  // group addresses sample 16kk + 8hg + 4hi + q, feature piece p (lane half hp = p&1, element half p>>1).
  // Inline asm, because the compiler orders every LDS read it knows of behind ALL outstanding LDS-DMA (vmcnt(0)),
  // which would serialise the staging ring; the reads are landed by land_ops() before their registers are used
  // (scripts/check_inflight.py verifies that on the final ISA).
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char*)smem;
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const uint32_t tr_off0 = (uint32_t)((tg & 1) * 1024 + wb_slot(tg & 1, tp & 1, 8 * (tg >> 1) + tq) * 16 + (tp >> 1) * 8);       // samples +0..3
  const uint32_t tr_off1 = (uint32_t)((tg & 1) * 1024 + wb_slot(tg & 1, tp & 1, 8 * (tg >> 1) + 4 + tq) * 16 + (tp >> 1) * 8);   // samples +4..7
  const uint32_t a_off0 = lds0 + tr_off0 + a_tile0 * 2048, a_off1 = lds0 + tr_off1 + a_tile0 * 2048;
  const uint32_t b_off0 = lds0 + tr_off0 + 16384 + b_tile0 * 2048, b_off1 = lds0 + tr_off1 + 16384 + b_tile0 * 2048;
  auto read_ops = [&](int c, int kk, WbRaw<AT>& a, WbRaw<BT>& b) {
    const uint32_t st = (uint32_t)(c - c0) % (uint32_t)kWbStages * kWbStageBytes + kk * 256;
    const uint32_t a0v = a_off0 + st, a1v = a_off1 + st, b0v = b_off0 + st, b1v = b_off1 + st;
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a.lo[i]) : "v"(a0v), "n"(i * 2048));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a.hi[i]) : "v"(a1v), "n"(i * 2048));
    }
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b.lo[j]) : "v"(b0v), "n"(j * 2048));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b.hi[j]) : "v"(b1v), "n"(j * 2048));
    }
  };
  auto land_ops = [&](WbRaw<AT>& ra, WbRaw<BT>& rb, WbOps<AT>& a, WbOps<BT>& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      asm volatile("" : "+v"(ra.lo[i]), "+v"(ra.hi[i]));
      const u32x4 t = {ra.lo[i][0], ra.lo[i][1], ra.hi[i][0], ra.hi[i][1]};
      a.v[i] = __builtin_bit_cast(bf16x8, t);
    }
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      asm volatile("" : "+v"(rb.lo[j]), "+v"(rb.hi[j]));
      const u32x4 t = {rb.lo[j][0], rb.lo[j][1], rb.hi[j][0], rb.hi[j][1]};
      b.v[j] = __builtin_bit_cast(bf16x8, t);
    }
  };
  const bf16x2 one2 = {(__bf16)1.0f, (__bf16)1.0f};
  auto mma = [&](const WbOps<AT>& a, const WbOps<BT>& b) {
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int j = 0; j < BT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[i], b.v[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < AT; ++i)
      if (do_bias[i]) {   // row sums of A: this lane's 8 samples of row li
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bf16x2 pr = {a.v[i][2 * e], a.v[i][2 * e + 1]};
          bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, one2, bsum[i], false);
        }
      }
  };

  // Software pipeline, in half-chunks of 16 samples: the MFMAs of one half run while the next half's operands are
  // fetched from LDS and half of the pieces of a chunk kWbAhead(+1) ahead are issued.  Mid-chunk: wait for chunk c+1
  // (all but the chunks behind it have landed), barrier, and the stage of chunk c-1 is free for c+kWbAhead+1.
  constexpr int PH = PW / 2;
#pragma unroll
  for (int r = 0; r < kWbAhead; ++r)
#pragma unroll
    for (int p = 0; p < PW; ++p) issue(p, c0 + r);
#pragma unroll
  for (int p = 0; p < PH; ++p) issue(p, c0 + kWbAhead);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kWbAhead - 1) * PW + PH) : "memory");
  __builtin_amdgcn_s_barrier();
  WbRaw<AT> ra0, ra1;
  WbRaw<BT> rb0, rb1;
  WbOps<AT> a0, a1;
  WbOps<BT> b0, b1;
  read_ops(c0, 0, ra0, rb0);
  land_ops(ra0, rb0, a0, b0);
  for (int c = c0; c < cl; ++c) {   // no read is in flight across the back edge (its registers could be copied)
    read_ops(c, 1, ra1, rb1);
#pragma unroll
    for (int p = PH; p < PW; ++p) issue(p, c + kWbAhead);
    mma(a0, b0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kWbAhead - 1) * PW) : "memory");   // chunk c+1 has landed
    __builtin_amdgcn_s_barrier();            // ... for every wave; every wave is done reading chunk c-1
    land_ops(ra1, rb1, a1, b1);
    read_ops(c + 1, 0, ra0, rb0);
#pragma unroll
    for (int p = 0; p < PH; ++p) issue(p, c + kWbAhead + 1);   // into the stage chunk c-1 used
    mma(a1, b1);
    land_ops(ra0, rb0, a0, b0);
  }
  read_ops(cl, 1, ra1, rb1);
  mma(a0, b0);
  land_ops(ra1, rb1, a1, b1);
  mma(a1, b1);
  // flush: fp32 atomics; rows/cols outside the product (tile padding, rows before a_row0) are dropped
#pragma unroll
  for (int i = 0; i < AT; ++i) {
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      const int col = 32 * (b_tile0 + j) + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * (a_tile0 + i) + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= it.a_row0 && row < it.a_rows && col < it.b_rows) {
          float* o = row < it.split ? it.dW + (int64_t)(row - it.a_row0) * it.ldo : it.dW2 + (int64_t)(row - it.split) * it.ldo;
          __hip_atomic_fetch_add((__attribute__((address_space(1))) float*)(o + it.col0 + col), acc[i][j][r], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (do_bias[i]) {   // both lane halves hold a partial sum of row li
      const int row = 32 * (a_tile0 + i) + li;
      if (row >= it.a_row0 && row < it.a_rows)
        __hip_atomic_fetch_add((__attribute__((address_space(1))) float*)(row < it.split ? it.db + row - it.a_row0 : it.db2 + row - it.split),
                               bsum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#ifdef WGB_STAMPS   // diagnostic build only (scripts/x3w_stamps.py <lib> bf16): cycles of every work entry, span of every workgroup
__device__ unsigned long long g_wgb_entry[8192 * 2];
__device__ unsigned long long g_wgb_span[1024 * 2];
#endif
__global__ __launch_bounds__(256, 1) void wgrad_bf16_kernel(const WgradBfItem* __restrict__ items,
                                                            const int32_t* __restrict__ first) {
  extern __shared__ __attribute__((aligned(1024))) char smem_b[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int e0 = first[blockIdx.x], e1 = first[blockIdx.x + 1];
#ifdef WGB_STAMPS
  if (threadIdx.x == 0) g_wgb_span[2 * blockIdx.x] = wall_clock64();
#endif
  for (int e = e0; e < e1; ++e) {
    const WgradBfItem it = items[e];
    if (it.c0 >= it.c1) continue;
#ifdef WGB_STAMPS
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    const int pieces = 2 * (((it.a_rows + 31) >> 5) + ((it.b_rows + 31) >> 5));
    switch (it.shape | (pieces > 16 ? 0x100 : 0)) {
      case 0x144: wgrad_bf16_body<4, 4, 8>(it, smem_b, wave, lane); break;
      case 0x142: wgrad_bf16_body<4, 2, 8>(it, smem_b, wave, lane); break;
      case 0x141: wgrad_bf16_body<4, 1, 8>(it, smem_b, wave, lane); break;
      case 0x124: wgrad_bf16_body<2, 4, 8>(it, smem_b, wave, lane); break;
      case 0x114: wgrad_bf16_body<1, 4, 8>(it, smem_b, wave, lane); break;
      case 0x112: wgrad_bf16_body<1, 2, 8>(it, smem_b, wave, lane); break;
      case 0x121: wgrad_bf16_body<2, 1, 8>(it, smem_b, wave, lane); break;
      case 0x042: wgrad_bf16_body<4, 2, 4>(it, smem_b, wave, lane); break;
      case 0x041: wgrad_bf16_body<4, 1, 4>(it, smem_b, wave, lane); break;
      case 0x024: wgrad_bf16_body<2, 4, 4>(it, smem_b, wave, lane); break;
      case 0x022: wgrad_bf16_body<2, 2, 4>(it, smem_b, wave, lane); break;
      case 0x021: wgrad_bf16_body<2, 1, 4>(it, smem_b, wave, lane); break;
      case 0x014: wgrad_bf16_body<1, 4, 4>(it, smem_b, wave, lane); break;
      case 0x012: wgrad_bf16_body<1, 2, 4>(it, smem_b, wave, lane); break;
      case 0x011: wgrad_bf16_body<1, 1, 4>(it, smem_b, wave, lane); break;
      default:    wgrad_bf16_body<2, 2, 8>(it, smem_b, wave, lane); break;   // 0x122 (and nothing else: see add_items_bf16)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every wave is done with the LDS stages before the next entry requests into them
#ifdef WGB_STAMPS
    if (threadIdx.x == 0) {
      g_wgb_entry[2 * e] = __builtin_amdgcn_s_memtime() - t0;
      g_wgb_entry[2 * e + 1] = ((unsigned long long)it.a_rows << 48) | ((unsigned long long)it.b_rows << 32) | (unsigned long long)(it.c1 - it.c0) << 12 | blockIdx.x;
    }
#endif
  }
#ifdef WGB_STAMPS
  if (threadIdx.x == 0) g_wgb_span[2 * blockIdx.x + 1] = wall_clock64();
#endif
}
#ifdef WGB_STAMPS
extern "C" int nerfhip_debug_wgb_stamps(unsigned long long* entry, unsigned long long* span) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(entry, HIP_SYMBOL(g_wgb_entry), sizeof(unsigned long long) * 8192 * 2) != hipSuccess) return 1;
  return hipMemcpyFromSymbol(span, HIP_SYMBOL(g_wgb_span), sizeof(unsigned long long) * 1024 * 2) != hipSuccess;
}
#endif

// positionCode(viewdir, dirL) of every sample's ray in operand layout (KT = 1): B operand of the Wd gradient
__global__ __launch_bounds__(256) void expand_encdir_bf16_kernel(const float* __restrict__ viewdir, int64_t M, int S, int Ed,
                                                                 uint16_t* __restrict__ out) {
  const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int64_t tix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t m = tix * 32 + li;
  const bool valid = m < M;
  const int64_t ray = (valid ? m : M - 1) / S;
  const float v[3] = {viewdir[ray * 3], viewdir[ray * 3 + 1], viewdir[ray * 3 + 2]};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = bf_feature(s, 0, e) + 4 * h;
      float x = 0.f;
      if (valid && f < Ed) {
        if (f < 3) x = v[f];
        else {
          const int k = f - 3, fr = k / 6, rem = k - 6 * fr, c = rem >= 3 ? rem - 3 : rem;
          const float a = v[c] * (float)(1 << fr);
          x = rem >= 3 ? cosf(a) : sinf(a);
        }
      }
      t[e] = (__bf16)x;
    }
    *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(out) + (tix * 2 + s) * 1024 + lane * 16) = t;
  }
}

// once per device, with that device current (nerfhip_create)
hipError_t prepare_wgrad_bf16() {
  return hipFuncSetAttribute((const void*)wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)((size_t)kWbStages * kWbStageBytes));
}

hipError_t launch_wgrad_bf16(const WgradBfItem* d_items, const int32_t* d_first, int ngroups, hipStream_t s) {
  if (ngroups <= 0) return hipSuccess;
  const size_t lds = (size_t)kWbStages * kWbStageBytes;   // the whole LDS (prepare_wgrad_bf16 raised the limit)
  hipLaunchKernelGGL(wgrad_bf16_kernel, dim3(ngroups), dim3(256), lds, s, d_items, d_first);
  return hipGetLastError();
}

// M: real samples; Mpad: samples covered by the store (tiles past M are written as zeros)
hipError_t launch_expand_encdir_bf16(const float* viewdir, int64_t M, int64_t Mpad, int S, int Ed, uint16_t* out, hipStream_t s) {
  const int64_t tiles = (Mpad + 31) / 32;
  hipLaunchKernelGGL(expand_encdir_bf16_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, viewdir, M, S, Ed, out);
  return hipGetLastError();
}

}  // namespace nh
