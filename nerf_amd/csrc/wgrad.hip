// wgrad.hip -- weight-gradient GEMM of the MLP on fp32 MFMA (gfx950), plus the small reductions around it.
//
//   dW[i][j] = sum_m  A[i][m] * B[j][m]          A = dY^T (gradient rows), B = X^T (input feature rows)
// Both operands are stored tile-major [Mpad/32][rows][32] by the forward/backward kernels (common.h), i.e.
// K(=sample)-contiguous for BOTH MFMA operands ("NT" GEMM with a huge reduction dimension and a <=256x256
// output) and one K-chunk of an operand is a single contiguous rows*128-byte block of HBM.  At the fp32 MFMA
// peak this kernel has to stream 64 flop/byte = 2.45 TB/s of operands, so the access pattern matters.
// One workgroup (4 waves, 1 per SIMD, 512-register file) owns a whole <=256x256 output tile in
// accumulators (a wave holds up to 4x4 MFMA tiles = 256 registers; the 4 waves are arranged 2x2, 1x4 or 4x1
// over the output by the host so that narrow outputs -- the 63-row encoding, the 128-row head -- still use
// all four SIMDs) and walks a slice of the sample axis
// in chunks of 32 samples staged by LDS-DMA into an XOR-swizzled image; slices of the sample axis are
// combined with fp32 atomics into the (pre-zeroed) canonical gradient buffer: per 256x256 layer that is
// (#slices) x 256 KiB of atomic traffic, two to three orders of magnitude below the operand stream.
// The kernel is persistent: n_cu workgroups, each working through an equal share of all products' sample axes
// (host: train_impl.inc), so there is no tail and only ~n_cu + #products accumulator flushes per step.
// The k order inside an 8-sample group is permuted (lane half h takes samples 4h..4h+3) identically for A
// and B, so one 16-byte LDS read feeds four MFMAs.
#include "mlp_common.hip.h"
#include <type_traits>
#include "launchers.h"

namespace nh {


constexpr int kWgRows = 256;                 // max rows per operand tile
constexpr int kWgChunkFloats = kWgRows * 32; // one operand, one chunk: 256 rows x 32 samples
constexpr int kWgBufBytes = 2 * kWgChunkFloats * 4;   // one buffer = A chunk | B chunk = 64 KiB

// LDS image of one operand chunk: [row][8 x 16-byte slots], slot s of row r holds samples
// 4*(s ^ swz(r)) .. +3 with swz(r) = (r >> 1) & 7  (conflict-free ds_read_b128 across 16 rows); staged by LDS-DMA
// with the swizzle applied on the source side (wgrad_body: offA / offB).

// Fragment reads are inline asm with counted lgkmcnt waits (the compiler would otherwise put every
// ds_read_b128 right in front of its MFMAs behind an lgkmcnt(0)): the AT+BT reads of k-group g+1 are issued
// before the AT*BT*4 MFMAs of group g.  LDS operations retire in order.
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
// fp32 add without return on a GLOBAL pointer (a generic pointer gives flat_atomic_add, which also ties up lgkmcnt)
__device__ __forceinline__ void gatomic_add(float* p, float v) {
  __hip_atomic_fetch_add((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N)); }

// One workgroup = one work entry; wave (wa, wb) owns A tiles [AT*wa, +AT) x B tiles [BT*wb, +BT).
// XR: the entry carries an extra vector product (WgradItem::x_row): row x_row of A's buffer against every B row, summed
// with fp32 FMAs on the B fragments the waves hold anyway (the density row behind the head gradient rows: one read
// of the feature buffer instead of two)
template <int AT, int BT, bool XR = false>
__device__ __forceinline__ void wgrad_body(const WgradItem& it, float* smem, int wave, int lane) {
  static_assert((AT == 1 || AT == 2 || AT == 4) && (BT == 1 || BT == 2 || BT == 4), "");
  const int li = lane & 31, h = lane >> 5;
  const int wa = wave / it.wb_n, wb = wave - wa * it.wb_n;
  const int a_row0 = 32 * AT * wa, b_row0 = 32 * BT * wb;
  const int c0 = it.c0, c1 = it.c1;   // this work entry's slice of the sample axis, in 32-sample chunks

  f32x16 acc[AT][BT];
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int j = 0; j < BT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias gradient = row sums of A: the waves with wb == 0 add up the A fragments they hold anyway (lane half h
  // has 4 of every 8 samples of row 32i + li; the halves are combined at the end) -- no extra LDS traffic
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

  // Staging: wave-instruction u of an operand covers rows 8u..8u+7 (64 lanes x 16 B); this wave issues u = wave + 4t,
  // t < 8.  The per-lane source offsets (row clamp + swizzle) are computed once; per chunk only the scalar base
  // moves.  The pieces of a chunk are NOT issued in one burst (an LDS-DMA instruction costs 60-185 issue cycles
  // next to LDS reads): part 0 goes behind the hand-over barrier, parts 1 and 2 into the next chunk's first two
  // k-groups, each among that group's MFMAs; all of them have landed at the next hand-over (vmcnt(0)).
  const int64_t a_step = (int64_t)it.a_stride * 32, b_step = (int64_t)it.b_stride * 32;   // floats per chunk
  const int a_stage = XR ? max(it.a_rows, it.x_row + 1) : it.a_rows;   // rows of A's buffer that are staged
  const int nA8 = (a_stage + 7) >> 3, nB8 = (it.b_rows + 7) >> 3;
  uint32_t offA[8], offB[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int r = 8 * (wave + 4 * t) + (lane >> 3), sl = (lane & 7) ^ ((r >> 1) & 7);
    offA[t] = (uint32_t)((min(r, a_stage - 1) * 32 + 4 * sl) * 4);   // rows past the end repeat the last row (never stored)
    offB[t] = (uint32_t)((min(r, it.b_rows - 1) * 32 + 4 * sl) * 4);
  }
  // 256x256-class products (AT = BT = 4, 7 of every 8.5 chunks): the pieces are issued BETWEEN the k-steps of the
  // MFMAs, branch-free (a branch inside the MFMA region makes the compiler copy registers that asm reads still
  // have in flight); other shapes issue theirs in front of the k-group's MFMAs.
  constexpr bool SPREAD = AT == 4 && BT == 4;
  auto stage_t = [&](int c, int buf, int t) {   // pieces u = wave + 4t of both operands of chunk c
    float* dst = smem + buf * 2 * kWgChunkFloats;
    const char* ga = reinterpret_cast<const char*>(it.A + c * a_step);
    const char* gb = reinterpret_cast<const char*>(it.B + c * b_step);
    const int u = wave + 4 * t;
    const uint32_t oa = t == 0 ? offA[0] : t == 1 ? offA[1] : t == 2 ? offA[2] : t == 3 ? offA[3] : t == 4 ? offA[4] : t == 5 ? offA[5] : t == 6 ? offA[6] : offA[7];
    const uint32_t ob = t == 0 ? offB[0] : t == 1 ? offB[1] : t == 2 ? offB[2] : t == 3 ? offB[3] : t == 4 ? offB[4] : t == 5 ? offB[5] : t == 6 ? offB[6] : offB[7];
    // SPREAD: no conditions (a piece past the operand's rows re-reads its last row into an image row nobody stores)
    if (SPREAD || u < nA8) __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(ga + oa), NERFHIP_LPTR(dst + u * 256), 16, 0, 0);
    if (SPREAD || u < nB8) __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(gb + ob), NERFHIP_LPTR(dst + kWgChunkFloats + u * 256), 16, 0, 0);
  };
  auto stage = [&](int c, int buf) {
#pragma unroll
    for (int t = 0; t < 8; ++t) stage_t(c, buf, t);
  };
  stage(c0, 0);
  if (c0 + 1 < c1) stage(c0 + 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 fa[2][AT], fb[2][BT];
  // XR: the k-group's four samples of the extra vector (the same for every lane of a half), per-B-row sums, its own sum
  constexpr int NR = AT + BT + (XR ? 1 : 0);   // LDS reads of one k-group
  f32x4 fe[2];
  float dsum[BT], esum = 0.f;
#pragma unroll
  for (int j = 0; j < BT; ++j) dsum[j] = 0.f;
  const uint32_t laneX = XR ? lds0 + (uint32_t)it.x_row * 128u : 0u;
  uint32_t xoffX[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xoffX[g] = XR ? 16u * ((uint32_t)(2 * g + h) ^ (uint32_t)((it.x_row >> 1) & 7)) : 0u;
  auto reads = [&](int set, uint32_t bufoff, int g) {
    if (set == 0) { read_frags<AT>(fa[0], laneA + bufoff + xoff[g]); read_frags<BT>(fb[0], laneB + bufoff + xoff[g]); }
    else          { read_frags<AT>(fa[1], laneA + bufoff + xoff[g]); read_frags<BT>(fb[1], laneB + bufoff + xoff[g]); }
    if constexpr (XR) {
      if (set == 0) lds_read16<0>(fe[0], laneX + bufoff + xoffX[g]);
      else          lds_read16<0>(fe[1], laneX + bufoff + xoffX[g]);
    }
  };
  // the 4 x AT x BT MFMAs of one k-group; SPREAD and PART >= 0: the staging pieces of part PART of chunk pc are
  // issued between the k-steps (part 0: t = 0,1; part 1: t = 2,3,4; part 2: t = 5,6,7)
  auto mfmas = [&](f32x4 (&a)[AT], f32x4 (&b)[BT], f32x4& e, auto part, int pc, int pbuf) {
    constexpr int PART = decltype(part)::value;
    if constexpr (XR) {
      esum += (e[0] + e[1]) + (e[2] + e[3]);
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        float d = dsum[j];
        d = fmaf(e[0], b[j][0], d); d = fmaf(e[1], b[j][1], d); d = fmaf(e[2], b[j][2], d); d = fmaf(e[3], b[j][3], d);
        dsum[j] = d;
      }
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < AT; ++i) bs[i] += (a[i][0] + a[i][1]) + (a[i][2] + a[i][3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int j = 0; j < BT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
      if constexpr (SPREAD) {   // pc is always a chunk of this entry (see chunk()): no condition needed
        if constexpr (PART == 0) { if (k == 0 || k == 2) stage_t(pc, pbuf, k / 2); }
        if constexpr (PART == 1) { if (k < 3) stage_t(pc, pbuf, 2 + k); }
        if constexpr (PART == 2) { if (k < 3) stage_t(pc, pbuf, 5 + k); }
      }
    }
    if constexpr (SPREAD && PART >= 0) {   // keep the scheduler from bunching the pieces: one per 8 MFMAs
      constexpr int NP = PART == 0 ? 4 : 6;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // VMEM (LDS-DMA)
      }
    }
  };
  auto burst = [&](auto part, bool go, int pc, int pbuf) {   // the same pieces, in front of the k-group (!SPREAD)
    constexpr int PART = decltype(part)::value;
    if constexpr (!SPREAD) {
      if (go) {
        if constexpr (PART == 0) { stage_t(pc, pbuf, 0); stage_t(pc, pbuf, 1); }
        if constexpr (PART == 1) { stage_t(pc, pbuf, 2); stage_t(pc, pbuf, 3); stage_t(pc, pbuf, 4); }
        if constexpr (PART == 2) { stage_t(pc, pbuf, 5); stage_t(pc, pbuf, 6); stage_t(pc, pbuf, 7); }
      }
    }
  };
  using P_ = std::integral_constant<int, -1>;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  // One chunk: k-groups 0..3.  With a chunk to follow, the hand-over to its buffer (DMA landed for everyone,
  // first fragment reads) sits between the last wait and the last MFMAs.
  auto chunk = [&](int c, bool has_next) {
    const uint32_t cur = (uint32_t)((c - c0) & 1) * (uint32_t)kWgBufBytes;
    // Chunk c+1's parts 1 and 2 go into the buffer chunk c-1 used (free since the last hand-over).  SPREAD issues
    // unconditionally: in the first chunk that re-stages chunk c0+1 (identical bytes over a buffer nobody reads
    // yet), past the end it stages the last chunk into the buffer nobody reads any more.
    const bool more = c > c0 && c + 1 < c1;
    const int nbuf = (c + 1 - c0) & 1;
    const int pc1 = min(c + 1, c1 - 1), pc2 = min(c + 2, c1 - 1);
    auto tie_e = [&](int set) { if constexpr (XR) { if (set == 0) asm volatile("" : "+v"(fe[0])); else asm volatile("" : "+v"(fe[1])); } };
    reads(1, cur, 1);
    burst(P1{}, more, c + 1, nbuf);
    wait_lgkm<NR>(); tie_frags<AT>(fa[0]); tie_frags<BT>(fb[0]); tie_e(0);
    mfmas(fa[0], fb[0], fe[0], P1{}, pc1, nbuf);
    reads(0, cur, 2);
    burst(P2{}, more, c + 1, nbuf);
    wait_lgkm<NR>(); tie_frags<AT>(fa[1]); tie_frags<BT>(fb[1]); tie_e(1);
    mfmas(fa[1], fb[1], fe[1], P2{}, pc1, nbuf);
    reads(1, cur, 3);
    wait_lgkm<NR>(); tie_frags<AT>(fa[0]); tie_frags<BT>(fb[0]); tie_e(0);
    mfmas(fa[0], fb[0], fe[0], P_{}, 0, 0);
    wait_lgkm<0>(); tie_frags<AT>(fa[1]); tie_frags<BT>(fb[1]); tie_e(1);
    if (has_next) {
      // every wave is done READING this buffer; chunk c+1 must have landed for all
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      burst(P0{}, c + 2 < c1, c + 2, (c - c0) & 1);
      reads(0, cur ^ (uint32_t)kWgBufBytes, 0);
      mfmas(fa[1], fb[1], fe[1], P0{}, pc2, (c - c0) & 1);   // ... all of it behind these MFMAs
      if constexpr (XR) {
        // land the next chunk's first reads (issued 32 MFMAs ago) in front of the back edge: with the extra fragment
        // live across it the compiler copies fragment registers on the loop's exit edge (scripts/check_inflight.py)
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<0>(); tie_frags<AT>(fa[0]); tie_frags<BT>(fb[0]); tie_e(0);
      }
    } else {
      mfmas(fa[1], fb[1], fe[1], P_{}, 0, 0);
    }
  };
  reads(0, 0u, 0);
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
        if (row < it.a_rows && col < it.b_rows) gatomic_add(it.dW + (int64_t)row * it.ldo + it.col0 + col, acc[i][j][r]);
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
  if constexpr (XR) {   // the two lane halves hold different samples of the same B row; the waves with wa == 0 flush
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

// Persistent workgroup b works through the entries first[b] .. first[b+1]-1 (host: train_impl.inc).
#ifdef WGF_STAMPS   // diagnostic build only (scripts/x3w_stamps.py <lib> f32): cycles of every work entry, span of every workgroup
__device__ unsigned long long g_wgf_entry[8192 * 2];
__device__ unsigned long long g_wgf_span[1024 * 2];
#endif
__global__ __launch_bounds__(256, 1) void wgrad_f32_kernel(const WgradItem* __restrict__ items,
                                                           const int32_t* __restrict__ first) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 buffers x (A chunk | B chunk)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int e0 = first[blockIdx.x], e1 = first[blockIdx.x + 1];
#ifdef WGF_STAMPS
  if (threadIdx.x == 0) g_wgf_span[2 * blockIdx.x] = wall_clock64();
#endif
  for (int e = e0; e < e1; ++e) {
    const WgradItem it = items[e];
    if (it.c0 >= it.c1) continue;
#ifdef WGF_STAMPS
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
#ifdef WGF_STAMPS
    if (threadIdx.x == 0) {
      g_wgf_entry[2 * e] = __builtin_amdgcn_s_memtime() - t0;
      g_wgf_entry[2 * e + 1] = ((unsigned long long)it.a_rows << 48) | ((unsigned long long)it.b_rows << 32) | (unsigned long long)(it.c1 - it.c0) << 12 | blockIdx.x;
    }
#endif
  }
#ifdef WGF_STAMPS
  if (threadIdx.x == 0) g_wgf_span[2 * blockIdx.x + 1] = wall_clock64();
#endif
}
#ifdef WGF_STAMPS
extern "C" int nerfhip_debug_wgf_stamps(unsigned long long* entry, unsigned long long* span) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(entry, HIP_SYMBOL(g_wgf_entry), sizeof(unsigned long long) * 8192 * 2) != hipSuccess) return 1;
  return hipMemcpyFromSymbol(span, HIP_SYMBOL(g_wgf_span), sizeof(unsigned long long) * 1024 * 2) != hipSuccess;
}
#endif

// d(dirvec)^T(j, ray) = sum over the ray's samples of dh(j, ray*S + s); both buffers tile-major (common.h).
// One workgroup per ray; a wave-load covers 8 rows x 32 samples of one tile (16 B per lane, 1 KiB per instruction),
// the 8 lanes of a row are combined with three DPP-able shuffles.
__global__ __launch_bounds__(256) void ray_sum_kernel(const float* __restrict__ dh, int dh_rows, int rows, int64_t n,
                                                      int S, float* __restrict__ out, int out_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ray = blockIdx.x;
  const int64_t m0 = ray * S, m1 = m0 + S;          // this ray's samples
  const int64_t t0 = m0 >> 5, t1 = (m1 - 1) >> 5;   // tiles they live in
  const int c4 = lane & 7, rsub = lane >> 3;
  for (int rg = wave; rg * 8 < rows; rg += 4) {
    const int row = rg * 8 + rsub;
    float s = 0.f;
    if (row < rows) {
      for (int64_t t = t0; t <= t1; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(dh + ((t * dh_rows + row) * 32 + 4 * c4));
        const int64_t m = t * 32 + 4 * c4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (m + e >= m0 && m + e < m1) s += v[e];
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (c4 == 0 && row < rows) out[tiled_index(row, ray, out_rows)] = s;
  }
}

// once per device, with that device current (nerfhip_create): allow the kernel its 128 KiB of dynamic LDS
hipError_t prepare_wgrad_f32() {
  return hipFuncSetAttribute((const void*)wgrad_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)((size_t)4 * kWgChunkFloats * sizeof(float)));
}

hipError_t launch_wgrad_f32(const WgradItem* d_items, const int32_t* d_first, int ngroups, hipStream_t s) {
  if (ngroups <= 0) return hipSuccess;
  const size_t lds = (size_t)4 * kWgChunkFloats * sizeof(float);  // 128 KiB (prepare_wgrad_f32 raised the limit)
  hipLaunchKernelGGL(wgrad_f32_kernel, dim3(ngroups), dim3(256), lds, s, d_items, d_first);
  return hipGetLastError();
}

hipError_t launch_ray_sum(const float* dh, int dh_rows, int rows, int64_t n, int S, float* out, int out_rows,
                          hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(ray_sum_kernel, dim3((unsigned)n), dim3(256), 0, s, dh, dh_rows, rows, n, S, out, out_rows);
  return hipGetLastError();
}

}  // namespace nh
