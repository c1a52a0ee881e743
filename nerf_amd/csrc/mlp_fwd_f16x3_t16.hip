// mlp_fwd_f16x3_t16.hip -- the INFERENCE forward of the split-f16 path on v_mfma_f32_16x16x32_f16 (gfx950).
//
// The arithmetic of mlp_fwd_f16x3.hip (f16x3_common.hip.h: every operand = hi + lo, three f16 MFMAs per product into one
// fp32 accumulator that carries the weights' factor 64; fp32-grade results, held to the fp32 parity tests) and its machinery
// (4 x 32 KiB LDS ring of (hi, lo) fragment pairs fed by LDS-DMA, pairs read ahead with counted waits, epilogue pieces
// between the next tile's MFMAs, persistent workgroups of 4 waves x 32 samples, one wave per SIMD) on the other MFMA shape,
// for the reason mlp_fwd_bf16_t16.hip gives: the render kernel is power-capped and the chip holds a higher clock under
// 16x16x32 MFMAs than under 32x32x16 ones (scripts/probe/mfma_shape2.hip: 1.10-1.20 x the FLOP/s on random data).  The
// training forward stays on mlp_fwd_f16x3.hip (its stores go out in the 32x32 accumulator layout).
//
// Tiling as in mlp_fwd_bf16_t16.hip: lane l = (c = l & 15, g = l >> 4); a wave's 32 samples are two halves n = 0, 1 of 16;
// output tile = 16 features, k-step = 32 features; register r of lane (c, g) of a finished tile = feature 16 ot + 4 g + r
// of sample (n, c), which is k-slot (g, j = 4 (ot & 1) + r) of k-step ot >> 1 of the next layer.  A fragment pair feeds SIX
// MFMAs (three terms x two sample halves).  The sin/cos of the encoding are the accurate ones (mlp_common.hip.h) and the
// encoding values are bit-identical to mlp_fwd_f16x3.hip's; what differs is the order in which an MFMA adds its products.
#include "f16x3_common.hip.h"
#include <algorithm>
#include "launchers.h"

namespace nh {
namespace x3 {
namespace t16 {

template <int NK>
struct ActP { Pair v[2][NK]; };   // [sample half][k-step of 32 features]: a (hi, lo) B operand pair each

template <int NK>
struct SelAct {   // k-steps 0 .. NK-1: previous layer; NK, NK+1: encoding (skip layers)
  const ActP<NK>& a;
  const ActP<2>* e;
  template <int KS_IDX> __device__ __forceinline__ const Pair& b(int n) const {
    if constexpr (KS_IDX < NK) return a.v[n][KS_IDX];
    else return e->v[n][KS_IDX - NK];
  }
};
struct SelEnc {
  const ActP<2>& e;
  template <int KS_IDX> __device__ __forceinline__ const Pair& b(int n) const { return e.v[n][KS_IDX]; }
};

// f16x3_common.hip.h: PartSched with ONE accumulator-init read per tile (4 rows per lane) instead of four
template <int KS, int OT, bool BIAS = true>
struct Sched {
  static constexpr int NF = KS * OT;
  static constexpr int kBiasStep = 1;
  static constexpr int pending(int j) {
    int pos = BIAS ? 1 : 0, need = (j == 0) ? pos : 0;
    for (int f = 0; f < (NF < kPD ? NF : kPD); ++f) { pos += 2; if (f == j) need = pos; }
    for (int t = 0; t <= j; ++t) {
      if (t + kPD < NF) { pos += 2; if (t + kPD == j && pos > need) need = pos; }
      if (BIAS && t % KS == kBiasStep && t / KS + 1 < OT) { pos += 1; if ((t / KS + 1) * KS == j && pos > need) need = pos; }
    }
    return pos - need;
  }
  static constexpr bool sync_after(int j) { return (j % kBlkPairs == kSyncPair) || (j == NF - 1 && (j % kBlkPairs) < kSyncPair); }
};

template <int KS, class Epi, int OT_IDX, int OFF, int P = 0>
struct Pieces {   // the pieces of tile OT_IDX scheduled at k-step offset OFF (OFF < 0: all); a = 64 x the tile, both halves
  static __device__ __forceinline__ void run(Epi& epi, const f32x4 (&a)[2]) {
    if constexpr (P < Epi::kPieces) {
      if constexpr (OFF < 0 || piece_offset<KS, Epi::kPieces>(P) == OFF) epi.template piece<OT_IDX, P>(a);
      Pieces<KS, Epi, OT_IDX, OFF, P + 1>::run(epi, a);
    }
  }
};

__device__ __forceinline__ f32x4 mfma16(const f16x8& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, as_f16x8(b), c, 0, 0, 0);
}

template <int KS, int OT, class BSel, class Epi, bool BIAS, int J>
struct Step {
  using S = Sched<KS, OT, BIAS>;
  static __device__ __forceinline__ void run(Ring& rg, int blk0, uint32_t bias_addr, f16x8 (&fr)[kPD + 1][2], f32x4& bias,
                                             f32x4 (&acc)[2][2], const BSel& bsel, Epi& epi) {
    constexpr int ot = J / KS, ks = J % KS;
    if constexpr (J + kPD < S::NF) {
      constexpr int jj = J + kPD;
      lds_read_frag<((2 * jj) % kBlkFrags) * 1024>(fr[jj % (kPD + 1)][0], rg.slot_addr(blk0 + jj / kBlkPairs));
      lds_read_frag<((2 * jj + 1) % kBlkFrags) * 1024>(fr[jj % (kPD + 1)][1], rg.slot_addr(blk0 + jj / kBlkPairs));
    }
    if constexpr (BIAS && ks == S::kBiasStep && ot + 1 < OT) lds_read_f4<16 * (ot + 1) * 4>(bias, bias_addr);
    wait_lgkm_n<(S::pending(J) < 15 ? S::pending(J) : 15)>();   // lgkmcnt is a 4-bit counter: a smaller count is only stricter
    asm volatile("" : "+v"(fr[J % (kPD + 1)][0]), "+v"(fr[J % (kPD + 1)][1]));
    const f16x8& wh = fr[J % (kPD + 1)][0];
    const f16x8& wl = fr[J % (kPD + 1)][1];
    const Pair& b0 = bsel.template b<ks>(0);
    const Pair& b1 = bsel.template b<ks>(1);
    f32x4 (&a)[2] = acc[ot & 1];
    if constexpr (ks == 0) {
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BIAS) { asm volatile("" : "+v"(bias)); c0 = bias; }
      a[0] = mfma16(wh, b0.hi, c0);
      a[1] = mfma16(wh, b1.hi, c0);
    } else {
      a[0] = mfma16(wh, b0.hi, a[0]);
      a[1] = mfma16(wh, b1.hi, a[1]);
    }
    a[0] = mfma16(wh, b0.lo, a[0]);
    a[1] = mfma16(wh, b1.lo, a[1]);
    a[0] = mfma16(wl, b0.hi, a[0]);
    a[1] = mfma16(wl, b1.hi, a[1]);
    if constexpr (ot >= 1) Pieces<KS, Epi, ot - 1, ks>::run(epi, acc[(ot - 1) & 1]);
    if constexpr (ot == OT - 1 && ks == KS - 1) Pieces<KS, Epi, ot, -1>::run(epi, acc[ot & 1]);
    if constexpr (S::sync_after(J)) { rg.blk = blk0 + J / kBlkPairs; rg.template sync_point<false>(); }
    if constexpr (J + 1 < S::NF) Step<KS, OT, BSel, Epi, BIAS, J + 1>::run(rg, blk0, bias_addr, fr, bias, acc, bsel, epi);
  }
};

// Runs one part that starts at block rg.blk (block-aligned) and leaves rg.blk at the next part's first block.
// bias_addr: LDS byte address of the part's accumulator-init vector + g*16.
template <int KS, int OT, bool BIAS = true, class BSel, class Epi>
__device__ __forceinline__ void run_part16(Ring& rg, uint32_t bias_addr, const BSel& bsel, Epi& epi) {
  static_assert(!BIAS || KS >= 2 || OT == 1, "the init rows of the next tile are requested at k-step 1");
  static_assert(kPD < kBlkPairs, "the prefetch must stay inside the block after the current one");
  using S = Sched<KS, OT, BIAS>;
  const int blk0 = rg.blk;
  f16x8 fr[kPD + 1][2];
  f32x4 bias;
  f32x4 acc[2][2];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (BIAS) lds_read_f4<0>(bias, bias_addr);
  prime_pairs<S::NF, 0>(fr, rg.slot_addr(blk0));
  Step<KS, OT, BSel, Epi, BIAS, 0>::run(rg, blk0, bias_addr, fr, bias, acc, bsel, epi);
  wait_lgkm_n<0>();
  __builtin_amdgcn_sched_barrier(0);
  rg.blk = blk0 + (S::NF + kBlkPairs - 1) / kBlkPairs;
}

// In every epilogue `a` is the finished accumulator tile = 64 x (W x + b) (f16x3_common.hip.h: kWScale), both sample halves.
__device__ __forceinline__ float relu_bits(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }
// range guard: mlp_fwd_f16x3.hip (an overflowed activation must not go on silently as zeros)
__device__ __forceinline__ uint32_t nonfinite_bits(float v) { return (__builtin_bit_cast(uint32_t, v) & 0x7f800000u) == 0x7f800000u ? 1u : 0u; }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void track_hi(uint32_t& mx, uint32_t h) {
  mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, mx), __builtin_bit_cast(u16x2, h)));
}
__device__ __forceinline__ uint32_t hi_overflowed(uint32_t mx) { return ((mx & 0xffffu) >= 0x7c00u || (mx >> 16) >= 0x7c00u) ? 1u : 0u; }

template <int NK>
struct EpiTrunk {   // relu + split -> half of a B operand pair of the next layer; piece P = (sample half P >> 1, registers 2 (P & 1), +1)
  ActP<NK>& out;
  uint32_t& mx;     // range guard: running maximum of the high halves made
  static constexpr int kPieces = 4;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x4 (&a)[2]) {
    constexpr int n = P >> 1, i = P & 1, w = 2 * (OT_IDX & 1) + i;
    split_scaled_into(out.v[n][OT_IDX >> 1], w, relu_bits(a[n][2 * i]), relu_bits(a[n][2 * i + 1]));
    track_hi(mx, out.v[n][OT_IDX >> 1].hi[w]);
  }
};
struct EpiSigma {   // one tile whose row 0 is the density row (nerfCoreBlock.scala:39-40)
  float* sigma; const int* m; const bool* valid; int g; uint32_t& bad;
  static constexpr int kPieces = 1;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x4 (&a)[2]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
      if (valid[n] && g == 0) { sigma[m[n]] = a[n][0] * kWInv; bad |= nonfinite_bits(a[n][0]); }
  }
};
template <int HT, int MODE>
struct EpiHead {    // tiles 0..2HT-1: first head matrix (16 rows each); tile 2HT: density row
  ActP<HT>& hid; float (&c)[2][3];
  const float* dirfeat; const int* ray; const int* m; const bool* valid; float* sigma; int g;
  uint32_t& bad;
  uint32_t& bad_mx;   // range guard: running maximum of the high halves made (colour hidden layer)
  float (&q)[2][2][4];   // MODE 2: [sample half][tile][register] = row 16 tile + 4 g + register of the SH coefficient matrix
  static constexpr int kPieces = 4;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x4 (&a)[2]) {
    constexpr int n = P >> 1, i = P & 1;
    if constexpr (OT_IDX == 2 * HT) {
      if constexpr (i == 0)
        if (valid[n] && g == 0) { sigma[m[n]] = a[n][0] * kWInv; bad |= nonfinite_bits(a[n][0]); }
    } else if constexpr (MODE == 1) {
      // add().relu() (nerfCoreBlock.scala:67): + (Wd.enc(dir) + bd + bf) of the sample's ray, fp32; registers 2i, 2i+1
      // hold features 16 OT + 4 g + 2 i + {0, 1}
      const f32x2 d = *reinterpret_cast<const f32x2*>(dirfeat + (int64_t)ray[n] * (32 * HT) + 16 * OT_IDX + 4 * g + 2 * i);
      const float v0 = fmaxf(fmaf(a[n][2 * i], kWInv, d[0]), 0.f), v1 = fmaxf(fmaf(a[n][2 * i + 1], kWInv, d[1]), 0.f);
      constexpr int w = 2 * (OT_IDX & 1) + i;
      split_into(hid.v[n][OT_IDX >> 1], w, v0, v1);
      track_hi(bad_mx, hid.v[n][OT_IDX >> 1].hi[w]);
    } else if constexpr (MODE == 2) {
      // the 27 SH coefficients (rows 0..26 of tiles 0 and 1) are only put aside here; sh_combine() multiplies them with
      // Y(dir) behind the part -- inside a part nothing but moves, so that the register allocator has no reason to shuffle
      // fragment registers whose loads are still in flight (scripts/check_inflight.py found exactly that at W = 256)
      q[n][OT_IDX][2 * i] = a[n][2 * i] * kWInv;
      q[n][OT_IDX][2 * i + 1] = a[n][2 * i + 1] * kWInv;
    } else {
      if constexpr (OT_IDX == 0) {
        if (g == 0) {
          if constexpr (i == 0) { c[n][0] = a[n][0] * kWInv; c[n][1] = a[n][1] * kWInv; }
          else c[n][2] = a[n][2] * kWInv;
        }
      }
    }
  }
};
struct EpiColour {   // colour output layer (rows 0..2 of its single tile)
  float (&c)[2][3]; int g;
  static constexpr int kPieces = 2;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x4 (&a)[2]) {
    if (g == 0) { c[P][0] = a[P][0] * kWInv; c[P][1] = a[P][1] * kWInv; c[P][2] = a[P][2] * kWInv; }
  }
};

// positionCode (nerfCoreBlock.scala:127-136) into the two B operand pairs of the encoding.  The element of k-slot (g, j)
// of k-step ks is feature e = 32 ks + 16 (j >> 2) + 4 g + (j & 3): it depends on the lane through g, so what an element is
// comes from a 1-KiB table in LDS, one float4 per (g, ks, j): (sx, sy, sz, flags) with exactly one of sx, sy, sz = 2^f
// (a one-hot select: x 2^f + 0 + 0 is exact, the argument is bit-identical to mlp_common.hip.h: code_value_h's) and
// flags = 1.0 for a cosine.  e < 3 (the coordinate itself: lanes g = 0, k-step 0, j < 3) is selected by a lane predicate;
// e >= E has a zero entry and encodes as sin(0) = 0.
__device__ __forceinline__ void enc_table_fill(f32x4* tab, int tid, int E) {
  if (tid < 64) {
    const int g = tid >> 4, idx = tid & 15, ks = idx >> 3, j = idx & 7;
    const int e = 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (e < E) {
      const CodeInfo ci = code_info(e);
      t[ci.coord] = ci.scale;
      t[3] = ci.want_cos ? 1.0f : 0.f;   // (a float: an integer bit pattern in a float register is a denormal, and not every move keeps those)
    }
    tab[g * 16 + idx] = t;
  }
}
template <bool RED>
__device__ __forceinline__ void encode(ActP<2>& enc, const float (&pos)[2][3], const f32x4* tab_g, int g, uint32_t& mx) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float v[2][2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * w + jj;
        const f32x4 t = tab_g[ks * 8 + j];
        const int wc = t[3] != 0.f ? 1 : 0;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          float a = pos[n][0] * t[0];
          a = __builtin_fmaf(pos[n][1], t[1], a);
          a = __builtin_fmaf(pos[n][2], t[2], a);
          float s;
          if constexpr (RED) {
            s = sin_or_cos_reduced(a, wc);
          } else {
            float sn, cs;
            sincosf(a, &sn, &cs);
            s = wc ? cs : sn;
          }
          if (ks == 0 && j < 3) s = (g == 0) ? a : s;   // e < 3: the coordinate itself
          v[n][jj] = s;
        }
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        split_into(enc.v[n][ks], w, v[n][0], v[n][1]);
        if (ks == 0 && w < 2) track_hi(mx, enc.v[n][ks].hi[w] & 0x7fff7fffu);   // the raw position: the only unbounded rows
      }
    }
  }
}

template <int WT, int MODE>
__global__ __launch_bounds__(64 * kWaves, 1) void mlp_fwd_f16x3_t16_kernel(NetArgs net, const char* __restrict__ stream,
                                                                           const float* __restrict__ bvec,
                                                                           const float* __restrict__ rays_o,
                                                                           const float* __restrict__ rays_d,
                                                                           const float* __restrict__ z,
                                                                           const float* __restrict__ dirfeat, int64_t M, int S,
                                                                           float* __restrict__ sigma, float* __restrict__ craw,
                                                                           int want_rgb, int nblk, int ntiles,
                                                                           uint32_t* __restrict__ nonfinite_flag) {
  constexpr int R = 32 * WT;
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring (kRing x 32 KiB) | vectors | encoding table
  float* vec = smem + kRing * kBlkBytes / 4;

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = net.D;
  const int nvec = D * R + (HT + 1) * 32 + 32;
  f32x4* tab = reinterpret_cast<f32x4*>(vec + ((nvec + 3) & ~3));

  Ring rg;
  rg.stream = stream;
  rg.ring = smem;
  rg.ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem + (uint32_t)lane * 16u;
  rg.wave = wave;
  rg.lane = lane;
  rg.voff = (uint32_t)(wave * (kBlkBytes / kWaves) + lane * 16);
  rg.nblk = nblk;
  for (int i = tid; i < nvec; i += 64 * kWaves) vec[i] = bvec[i];
  enc_table_fill(tab, tid, net.E);
  const uint32_t vec_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)vec + (uint32_t)g * 16u;
  const f32x4* tab_g = tab + g * 16;

  __syncthreads();   // vectors and table visible
  rg.start();

  uint32_t bad = 0u, mx = 0u;   // range guard: this lane wrote a non-finite output / running maximum of the f16 high halves made
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int m[2], ray[2];   // host guarantees M < 2^31
    bool valid[2];
    float pos[2][3];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      m[n] = tile * kTileSamples + wave * 32 + n * 16 + c;
      valid[n] = m[n] < (int)M;
      const int mc = valid[n] ? m[n] : (int)M - 1;
      ray[n] = (int)((uint32_t)mc / (uint32_t)S);
      // sample position (raysO.add(raysD.mul(zVals)), nerfBlock.scala:65): separate mul and add, fp32
      const float zz = z[mc];
#pragma unroll
      for (int k = 0; k < 3; ++k) pos[n][k] = __fadd_rn(rays_o[(int64_t)ray[n] * 3 + k], __fmul_rn(rays_d[(int64_t)ray[n] * 3 + k], zz));
    }
    // the encoding is computed once per sample and kept (32 registers) for the skip layers: its sin/cos are the
    // accurate ones, far too expensive to recompute
    ActP<2> enc;
    {
      const bool red = trig_reduced(pos[0][0], pos[0][1], pos[0][2], net.E) && trig_reduced(pos[1][0], pos[1][1], pos[1][2], net.E);
      if (red) encode<true>(enc, pos, tab_g, g, mx);   // wave-uniform: the usual case
      else encode<false>(enc, pos, tab_g, g, mx);
    }
    ActP<WT> act, nxt;
    {
      EpiTrunk<WT> epi{act, mx};
      run_part16<2, 2 * WT>(rg, vec_addr, SelEnc{enc}, epi);   // layer 0
    }
    auto layer = [&](int l, const ActP<WT>& in, ActP<WT>& out) __attribute__((always_inline)) {
      EpiTrunk<WT> epi{out, mx};
      const uint32_t baddr = vec_addr + (uint32_t)(l * R) * 4u;
      if ((net.skip_mask >> (l - 1)) & 1u) run_part16<WT + 2, 2 * WT>(rg, baddr, SelAct<WT>{in, &enc}, epi);
      else run_part16<WT, 2 * WT>(rg, baddr, SelAct<WT>{in, nullptr}, epi);
    };
    int l = 1;
    for (; l + 1 < D; l += 2) {
      layer(l, act, nxt);
      layer(l + 1, nxt, act);
    }
    if (l < D) {
      layer(l, act, nxt);
      act = nxt;
    }

    const uint32_t haddr = vec_addr + (uint32_t)(D * R) * 4u;
    if (!want_rgb) {
      // sigma-only stream: one tile, row 0 = density row (nerfCoreBlock.scala:39-40)
      EpiSigma epi{sigma, m, valid, g, bad};
      run_part16<WT, 1>(rg, haddr + (uint32_t)(HT * 32) * 4u, SelAct<WT>{act, nullptr}, epi);
      continue;
    }
    // head: 2 HT colour tiles, then the density tile
    ActP<HT> hid;
    float cl[2][3];
#pragma unroll
    for (int n = 0; n < 2; ++n) cl[n][0] = cl[n][1] = cl[n][2] = 0.f;
    float q[2][2][4];
    EpiHead<HT, MODE> epih{hid, cl, dirfeat, ray, m, valid, sigma, g, bad, mx, q};
    run_part16<WT, 2 * HT + 1>(rg, haddr, SelAct<WT>{act, nullptr}, epih);
    if constexpr (MODE == 1) {
      // colour output layer: Linear(3) over the H2 hidden features, as one MFMA tile (rows 0..2)
      EpiColour epic{cl, g};
      run_part16<HT, 1>(rg, haddr + (uint32_t)((HT + 1) * 32) * 4u, SelAct<HT>{hid, nullptr}, epic);
    } else if constexpr (MODE == 2) {
      // reshape(...,9,3).mul(SH).sum(-2) (nerfCoreBlock.scala:55): c[k] = sum_j q[3j+k] * Y_j(dir), rows 0..26
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const float* Y = dirfeat + (int64_t)ray[n] * 12;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * g + r;
            if (row < 27) {
              const int j = row / 3, k = row - 3 * j;
              const float v = q[n][t][r] * Y[j];
              cl[n][0] += k == 0 ? v : 0.f; cl[n][1] += k == 1 ? v : 0.f; cl[n][2] += k == 2 ? v : 0.f;
            }
          }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          cl[n][k] += __shfl_xor(cl[n][k], 16);
          cl[n][k] += __shfl_xor(cl[n][k], 32);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
      if (valid[n] && g == 0) {
        craw[(int64_t)m[n] * 3 + 0] = cl[n][0]; craw[(int64_t)m[n] * 3 + 1] = cl[n][1]; craw[(int64_t)m[n] * 3 + 2] = cl[n][2];
        bad |= nonfinite_bits(cl[n][0]) | nonfinite_bits(cl[n][1]) | nonfinite_bits(cl[n][2]);
      }
  }   // tile loop
  bad |= hi_overflowed(mx);
  if (bad && nonfinite_flag) atomicOr(nonfinite_flag, 1u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

template <int WT, int MODE>
static hipError_t launch_one(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o, const float* rays_d,
                             const float* z, const float* dirfeat, int64_t M, int S, float* sigma, float* craw, int want_rgb,
                             int nblk, int n_cu, uint32_t* nonfinite_flag, hipStream_t stream) {
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  const size_t nvec = (size_t)net.D * 32 * WT + (HT + 1) * 32 + 32;
  const size_t lds = (size_t)kRing * kBlkBytes + ((nvec + 3) & ~(size_t)3) * sizeof(float) + 64 * sizeof(f32x4);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int ntiles = (int)((M + kTileSamples - 1) / kTileSamples);
  const dim3 grid((unsigned)std::min(ntiles, n_cu)), block(64 * kWaves);   // one persistent workgroup per CU
  hipLaunchKernelGGL((mlp_fwd_f16x3_t16_kernel<WT, MODE>), grid, block, lds, stream, net, (const char*)stream_w, bvec, rays_o,
                     rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, ntiles, nonfinite_flag);
  return hipGetLastError();
}

}  // namespace t16
}  // namespace x3

#define NH_X3T_CASES(F)                                                                                                 \
  switch ((W / 32) * 4 + mode) {                                                                                        \
    case 2 * 4 + 0: F(2, 0); case 2 * 4 + 1: F(2, 1); case 2 * 4 + 2: F(2, 2);                                          \
    case 4 * 4 + 0: F(4, 0); case 4 * 4 + 1: F(4, 1); case 4 * 4 + 2: F(4, 2);                                          \
    case 8 * 4 + 0: F(8, 0); case 8 * 4 + 1: F(8, 1); case 8 * 4 + 2: F(8, 2);                                          \
    default: return hipErrorInvalidValue;                                                                               \
  }

// once per device, with that device current (nerfhip_create)
hipError_t prepare_mlp_fwd_f16x3_t16(int W, int mode) {
#define NH_X3T_PREP(WT_, MODE_)                                                                                        \
  return hipFuncSetAttribute((const void*)x3::t16::mlp_fwd_f16x3_t16_kernel<WT_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                             160 * 1024)
  NH_X3T_CASES(NH_X3T_PREP)
#undef NH_X3T_PREP
}

// stream_w: the 16x16x32-ordered full pair stream when want_rgb, else the density-only one (api.hip: build_fwd_streams_t16)
hipError_t launch_mlp_fwd_f16x3_t16(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                                    const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                                    float* craw, int want_rgb, int nblk, int n_cu, uint32_t* nonfinite_flag, hipStream_t stream) {
  const int W = net.W, mode = net.mode;
#define NH_X3T_LAUNCH(WT_, MODE_) \
  return x3::t16::launch_one<WT_, MODE_>(net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, nonfinite_flag, stream)
  NH_X3T_CASES(NH_X3T_LAUNCH)
#undef NH_X3T_LAUNCH
}

}  // namespace nh
