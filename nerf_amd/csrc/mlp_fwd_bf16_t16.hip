// mlp_fwd_bf16_t16.hip -- the INFERENCE forward of the bf16 path on v_mfma_f32_16x16x32_bf16 (gfx950).
//
// Same function, same machinery (bf16_common.hip.h: LDS ring fed by LDS-DMA, fragments read ahead with counted waits,
// epilogue pieces between the next tile's MFMAs, persistent workgroups of 8 waves x 32 samples) as mlp_fwd_bf16.hip, on the
// other MFMA shape.  Why: the bf16 render kernel is power-capped -- the chip lowers its clock under the matrix pipe's load --
// and the clock it holds depends on the shape: on random operands the bare 16x16x32 loop delivers 1.17-1.20 x the FLOP/s of
// the 32x32x16 loop with operands in registers and 1.10-1.12 x with the A operand re-read from LDS per MFMA group, at equal
// rates on zeros (scripts/probe/mfma_shape2.hip, MI355X_MICROARCH.md "DVFS give-back" item 7).  Training keeps the 32x32x16
// kernels (their activation store is laid out for the weight-gradient kernel's transposed reads).
//
// Tiling.  Lane l = (c = l & 15, g = l >> 4).  A wave owns 32 samples as two halves n = 0, 1 of 16 (sample = 16 n + c);
// one MFMA = 16 output features x 16 samples x 32 input features:
//   A (weights, from LDS): lane (c, g) holds W[16 ot + c][k-slots 8 g .. 8 g + 7] of the k-step      (1 KiB per fragment)
//   B (activations, in registers): lane (c, g) holds the same 8 k-slots of sample (n, c)
//   D: register r of lane (c, g) = output feature 16 ot + 4 g + r of sample (n, c)
// so the four registers of output tile ot are -- after ReLU and rounding -- half of a B operand of the next layer with no
// lane movement: k-step ks of the next layer takes tiles 2 ks (elements 0..3) and 2 ks + 1 (elements 4..7), i.e. k-slot
// (g, j) of k-step ks holds feature 32 ks + 16 (j >> 2) + 4 g + (j & 3); the weight fragments are packed in that order
// (pack.hip: FragDesc::transposed == 2).  Every fragment feeds TWO MFMAs (both sample halves): the same 1 KiB of LDS per
// 32 768 FLOP as the 32x32x16 kernel.
#include "bf16_common.hip.h"
#include <algorithm>
#include "launchers.h"

namespace nh {
namespace t16 {

using C = BfCfg<1, 8>;

template <int NK>
struct ActW { u32x4 v[2][NK]; };   // [sample half][k-step of 32 features]: a B operand each

template <int NK>
struct SelAct {   // k-steps 0 .. NK-1: previous layer; NK, NK+1: encoding (skip layers)
  const ActW<NK>& a;
  const ActW<2>* e;
  template <int KS_IDX> __device__ __forceinline__ const u32x4& b(int n) const {
    if constexpr (KS_IDX < NK) return a.v[n][KS_IDX];
    else return e->v[n][KS_IDX - NK];
  }
};
struct SelEnc {
  const ActW<2>& e;
  template <int KS_IDX> __device__ __forceinline__ const u32x4& b(int n) const { return e.v[n][KS_IDX]; }
};

// ---- one layer part: OT output tiles (16 rows) x KS k-steps (32 features), fragments j = ot*KS + ks -----------------
// bf16_common.hip.h: PartSched with ONE accumulator-init read per tile (4 rows per lane) instead of four
template <int KS, int OT, bool BIAS = true>
struct Sched {
  static constexpr int NF = KS * OT;
  static constexpr int kBiasStep = 1;
  static constexpr int pending(int j) {
    int pos = BIAS ? 1 : 0, need = (j == 0) ? pos : 0;
    for (int f = 0; f < (NF < kPD ? NF : kPD); ++f) { pos += 1; if (f == j) need = pos; }
    for (int t = 0; t <= j; ++t) {
      if (t + kPD < NF) { pos += 1; if (t + kPD == j && pos > need) need = pos; }
      if (BIAS && t % KS == kBiasStep && t / KS + 1 < OT) { pos += 1; if ((t / KS + 1) * KS == j && pos > need) need = pos; }
    }
    return pos - need;
  }
  static constexpr bool sync_after(int j) { return (j % kBlkFrags == kSyncFrag) || (j == NF - 1 && (j % kBlkFrags) < kSyncFrag); }
};

template <int KS, class Epi, int OT_IDX, int OFF, int P = 0>
struct Pieces {   // the pieces of tile OT_IDX scheduled at k-step offset OFF (OFF < 0: all)
  static __device__ __forceinline__ void run(Epi& epi, f32x4 (&acc)[2]) {
    if constexpr (P < Epi::kPieces) {
      if constexpr (OFF < 0 || piece_offset<KS, Epi::kPieces>(P) == OFF) epi.template piece<OT_IDX, P>(acc);
      Pieces<KS, Epi, OT_IDX, OFF, P + 1>::run(epi, acc);
    }
  }
};

__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int KS, int OT, class BSel, class Epi, bool BIAS, int J>
struct Step {
  using S = Sched<KS, OT, BIAS>;
  static __device__ __forceinline__ void run(BfRingT<C>& rg, int blk0, uint32_t bias_addr, bf16x8 (&fr)[kPD + 1], f32x4& bias,
                                             f32x4 (&acc)[2][2], const BSel& bsel, Epi& epi) {
    constexpr int ot = J / KS, ks = J % KS;
    if constexpr (J + kPD < S::NF) {
      constexpr int jj = J + kPD;
      lds_read_frag<(jj % kBlkFrags) * 1024>(fr[jj % (kPD + 1)], rg.slot_addr(blk0 + jj / kBlkFrags));
    }
    if constexpr (BIAS && ks == S::kBiasStep && ot + 1 < OT) lds_read_f4<16 * (ot + 1) * 4>(bias, bias_addr);
    wait_lgkm_n<S::pending(J)>();
    asm volatile("" : "+v"(fr[J % (kPD + 1)]));
    if constexpr (ks == 0) {
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BIAS) { asm volatile("" : "+v"(bias)); c0 = bias; }
      acc[ot & 1][0] = mfma16(fr[J % (kPD + 1)], bsel.template b<ks>(0), c0);
      acc[ot & 1][1] = mfma16(fr[J % (kPD + 1)], bsel.template b<ks>(1), c0);
    } else {
      acc[ot & 1][0] = mfma16(fr[J % (kPD + 1)], bsel.template b<ks>(0), acc[ot & 1][0]);
      acc[ot & 1][1] = mfma16(fr[J % (kPD + 1)], bsel.template b<ks>(1), acc[ot & 1][1]);
    }
    if constexpr (ot >= 1) Pieces<KS, Epi, ot - 1, ks>::run(epi, acc[(ot - 1) & 1]);
    if constexpr (ot == OT - 1 && ks == KS - 1) Pieces<KS, Epi, ot, -1>::run(epi, acc[ot & 1]);
    if constexpr (S::sync_after(J)) { rg.blk = blk0 + J / kBlkFrags; rg.sync_point(); }
    if constexpr (J + 1 < S::NF) Step<KS, OT, BSel, Epi, BIAS, J + 1>::run(rg, blk0, bias_addr, fr, bias, acc, bsel, epi);
  }
};

// Runs one part that starts at block rg.blk (block-aligned) and leaves rg.blk at the next part's first block.
// bias_addr: LDS byte address of the part's accumulator-init vector + g*16.
template <int KS, int OT, bool BIAS = true, class BSel, class Epi>
__device__ __forceinline__ void run_part(BfRingT<C>& rg, uint32_t bias_addr, const BSel& bsel, Epi& epi) {
  static_assert(!BIAS || KS >= 2 || OT == 1, "the init rows of the next tile are requested at k-step 1");
  static_assert(OT == 1 || piece_offset<KS, Epi::kPieces>(Epi::kPieces - 1) < KS, "epilogue pieces must fit into the next tile");
  using S = Sched<KS, OT, BIAS>;
  const int blk0 = rg.blk;
  bf16x8 fr[kPD + 1];
  f32x4 bias;
  f32x4 acc[2][2];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (BIAS) lds_read_f4<0>(bias, bias_addr);
  prime_frags_bf<S::NF, 0>(fr, rg.slot_addr(blk0));
  Step<KS, OT, BSel, Epi, BIAS, 0>::run(rg, blk0, bias_addr, fr, bias, acc, bsel, epi);
  wait_lgkm_n<0>();
  __builtin_amdgcn_sched_barrier(0);
  rg.blk = blk0 + (S::NF + kBlkFrags - 1) / kBlkFrags;
}

// two fp32 -> one packed bf16 pair (RNE), ReLU on the pair by v_pk_max_i16 (bf16_common.hip.h: relu_pack_half)
__device__ __forceinline__ unsigned relu_pair(float a, float b) {
  const f32x2 v = {a, b};
  const bf16x2 p = __builtin_convertvector(v, bf16x2);
  const i16x2 zero = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, p), zero));
}
__device__ __forceinline__ unsigned pack_pair(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <int NK>
struct EpiTrunk {   // relu + bf16 -> half of a B operand of the next layer
  ActW<NK>& out;
  static constexpr int kPieces = 2;   // sample half
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x4 (&acc)[2]) {
    out.v[P][OT_IDX >> 1][2 * (OT_IDX & 1)] = relu_pair(acc[P][0], acc[P][1]);
    out.v[P][OT_IDX >> 1][2 * (OT_IDX & 1) + 1] = relu_pair(acc[P][2], acc[P][3]);
  }
};
struct EpiSigma {   // one tile whose row 0 is the density row (nerfCoreBlock.scala:39-40)
  float* sigma; const int* m; const bool* valid; int g;
  static constexpr int kPieces = 1;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x4 (&acc)[2]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
      if (valid[n] && g == 0) sigma[m[n]] = acc[n][0];
  }
};
template <int HT, int MODE>
struct EpiHead {    // tiles 0..2HT-1: first head matrix (16 rows each); tile 2HT: density row
  ActW<HT>& hid; float (&c)[2][3];
  const float* dirfeat; const int* ray; const int* m; const bool* valid; float* sigma; int g;
  static constexpr int kPieces = 1;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x4 (&acc)[2]) {
    if constexpr (OT_IDX == 2 * HT) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
        if (valid[n] && g == 0) sigma[m[n]] = acc[n][0];
    } else if constexpr (MODE == 1) {
      // add().relu() (nerfCoreBlock.scala:67): + (Wd.enc(dir) + bd + bf) of the sample's ray, fp32
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dirfeat + (int64_t)ray[n] * (32 * HT) + 16 * OT_IDX + 4 * g);
        hid.v[n][OT_IDX >> 1][2 * (OT_IDX & 1)] = relu_pair(acc[n][0] + d[0], acc[n][1] + d[1]);
        hid.v[n][OT_IDX >> 1][2 * (OT_IDX & 1) + 1] = relu_pair(acc[n][2] + d[2], acc[n][3] + d[3]);
      }
    } else if constexpr (MODE == 2) {
      // reshape(...,9,3).mul(SH).sum(-2) (:55): c[k] = sum_j q[3j+k] * Y_j(dir); rows 0..26 (tiles 0 and 1)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const float* Y = dirfeat + (int64_t)ray[n] * 12;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * OT_IDX + 4 * g + r;
          if (row < 27) {
            const int j = row / 3, k = row - 3 * j;
            const float v = acc[n][r] * Y[j];
            c[n][0] += k == 0 ? v : 0.f; c[n][1] += k == 1 ? v : 0.f; c[n][2] += k == 2 ? v : 0.f;
          }
        }
      }
    } else {
      if constexpr (OT_IDX == 0) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
          if (g == 0) { c[n][0] = acc[n][0]; c[n][1] = acc[n][1]; c[n][2] = acc[n][2]; }
      }
    }
  }
};
struct EpiColour {   // colour output layer (rows 0..2 of its single tile)
  float (&c)[2][3]; int g;
  static constexpr int kPieces = 1;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x4 (&acc)[2]) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
      if (g == 0) { c[n][0] = acc[n][0]; c[n][1] = acc[n][1]; c[n][2] = acc[n][2]; }
  }
};

// positionCode (nerfCoreBlock.scala:127-136) straight into the two B operands of the encoding: the element of k-slot
// (g, j) of k-step ks is feature e = 32 ks + 16 (j >> 2) + 4 g + (j & 3), which depends on the lane through g, so what an
// element is -- which coordinate, which frequency, sin or cos -- comes from a 1-KiB table in LDS, one float4 per (g, ks, j):
// (sx, sy, sz, phase) with exactly one of sx, sy, sz = 2^f / 2 pi (the others 0: a one-hot select that is exact) and
// phase = 1/4 revolution for a cosine; value = v_sin(fract(x s) + phase).  The argument x 2^f / 2 pi is rounded exactly as
// in bf16_common.hip.h: code_value_fast (a power-of-two factor commutes with the rounding).  e < 3 (the coordinate itself:
// lanes g = 0, k-step 0, j < 3) is selected by a lane predicate; e >= E has a zero entry and encodes as sin(0) = 0.
__device__ __forceinline__ void enc_table_fill(f32x4* tab, int tid, int E) {
  if (tid < 64) {
    const int g = tid >> 4, idx = tid & 15, ks = idx >> 3, j = idx & 7;
    const int e = 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (e < E) {
      if (e < 3) {
        t[e] = 1.0f;
      } else {
        const int u = e - 3, f = u / 6, rem = u - 6 * f;
        const int cd = rem >= 3 ? rem - 3 : rem;
        t[cd] = (float)(1 << f) * 0.15915494309189535f;
        t[3] = rem >= 3 ? 0.25f : 0.f;
      }
    }
    tab[g * 16 + idx] = t;
  }
}
__device__ __forceinline__ void encode(ActW<2>& enc, const float (&pos)[2][3], const f32x4* tab_g, int g) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float v[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 t = tab_g[ks * 8 + j];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float r = pos[n][0] * t[0];
        r = __builtin_fmaf(pos[n][1], t[1], r);
        r = __builtin_fmaf(pos[n][2], t[2], r);
        float s = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r) + t[3]);
        if (ks == 0 && j < 3) s = (g == 0) ? r : s;
        v[n][j] = s;
      }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int w = 0; w < 4; ++w) enc.v[n][ks][w] = pack_pair(v[n][2 * w], v[n][2 * w + 1]);
  }
}

template <int WT, int MODE>
__global__ __launch_bounds__(512) void mlp_fwd_bf16_t16_kernel(NetArgs net, const char* __restrict__ stream,
                                                                const float* __restrict__ bvec,
                                                                const float* __restrict__ rays_o,
                                                                const float* __restrict__ rays_d,
                                                                const float* __restrict__ z,
                                                                const float* __restrict__ dirfeat, int64_t M, int S,
                                                                float* __restrict__ sigma, float* __restrict__ craw,
                                                                int want_rgb, int nblk, int ntiles) {
  constexpr int R = 32 * WT;
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring (kRing x 32 KiB) | vectors | encoding table
  float* vec = smem + kRing * kBlkBytes / 4;

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = net.D;
  const int nvec = D * R + (HT + 1) * 32 + 32;
  f32x4* tab = reinterpret_cast<f32x4*>(vec + ((nvec + 3) & ~3));

  BfRingT<C> rg;
  rg.stream = stream;
  rg.ring = smem;
  rg.ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem + (uint32_t)lane * 16u;
  rg.wave = wave;
  rg.lane = lane;
  rg.nblk = nblk;
  for (int i = tid; i < nvec; i += 64 * C::WAVES) vec[i] = bvec[i];
  enc_table_fill(tab, tid, net.E);
  const uint32_t vec_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)vec + (uint32_t)g * 16u;
  const f32x4* tab_g = tab + g * 16;

  __syncthreads();   // vectors and table visible
  rg.start();

  // Persistent: the workgroup walks sample tiles blockIdx.x, +gridDim.x, ...; the weight ring never stops (the
  // stream is consumed cyclically), so only the first tile pays its start-up latency.
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  int m[2], ray[2];   // host guarantees M < 2^31
  bool valid[2];
  float pos[2][3];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    m[n] = tile * C::TILE + wave * 32 + n * 16 + c;
    valid[n] = m[n] < (int)M;
    const int mc = valid[n] ? m[n] : (int)M - 1;
    ray[n] = (int)((uint32_t)mc / (uint32_t)S);
    // sample position (raysO.add(raysD.mul(zVals)), nerfBlock.scala:65): separate mul and add, fp32
    const float zz = z[mc];
#pragma unroll
    for (int k = 0; k < 3; ++k) pos[n][k] = __fadd_rn(rays_o[(int64_t)ray[n] * 3 + k], __fmul_rn(rays_d[(int64_t)ray[n] * 3 + k], zz));
  }
  ActW<WT> act, nxt;
  {
    ActW<2> enc;
    encode(enc, pos, tab_g, g);
    EpiTrunk<WT> epi{act};
    run_part<2, 2 * WT>(rg, vec_addr, SelEnc{enc}, epi);   // layer 0
  }
  // layers 1..D-1, two per trip (act -> nxt -> act) so the activations never have to be copied
  auto layer = [&](int l, const ActW<WT>& in, ActW<WT>& out) {
    EpiTrunk<WT> epi{out};
    const uint32_t baddr = vec_addr + (uint32_t)(l * R) * 4u;
    if ((net.skip_mask >> (l - 1)) & 1u) {
      // the encoding is recomputed HERE (opaque positions: not hoisted out of the layer loop, where it would be live --
      // and spilled -- through the whole trunk)
#pragma unroll
      for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(pos[n][0]), "+v"(pos[n][1]), "+v"(pos[n][2]));
      ActW<2> enc;
      encode(enc, pos, tab_g, g);
      run_part<WT + 2, 2 * WT>(rg, baddr, SelAct<WT>{in, &enc}, epi);
    } else {
      run_part<WT, 2 * WT>(rg, baddr, SelAct<WT>{in, nullptr}, epi);
    }
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
    EpiSigma epi{sigma, m, valid, g};
    run_part<WT, 1>(rg, haddr + (uint32_t)(HT * 32) * 4u, SelAct<WT>{act, nullptr}, epi);
    continue;
  }

  // head: 2 HT colour tiles, then the density tile
  ActW<HT> hid;
  float cl[2][3];
#pragma unroll
  for (int n = 0; n < 2; ++n) cl[n][0] = cl[n][1] = cl[n][2] = 0.f;
  EpiHead<HT, MODE> epih{hid, cl, dirfeat, ray, m, valid, sigma, g};
  run_part<WT, 2 * HT + 1>(rg, haddr, SelAct<WT>{act, nullptr}, epih);

  if constexpr (MODE == 1) {
    // colour output layer: Linear(3) over the H2 hidden features, as one MFMA tile (rows 0..2)
    EpiColour epic{cl, g};
    run_part<HT, 1>(rg, haddr + (uint32_t)((HT + 1) * 32) * 4u, SelAct<HT>{hid, nullptr}, epic);
  } else if constexpr (MODE == 2) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        cl[n][k] += __shfl_xor(cl[n][k], 16);
        cl[n][k] += __shfl_xor(cl[n][k], 32);
      }
  }
#pragma unroll
  for (int n = 0; n < 2; ++n)
    if (valid[n] && g == 0) { craw[(int64_t)m[n] * 3 + 0] = cl[n][0]; craw[(int64_t)m[n] * 3 + 1] = cl[n][1]; craw[(int64_t)m[n] * 3 + 2] = cl[n][2]; }
  }   // tile loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

template <int WT, int MODE>
static size_t lds_bytes(const NetArgs& net) {
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  const size_t nvec = (size_t)net.D * 32 * WT + (HT + 1) * 32 + 32;
  return (size_t)kRing * kBlkBytes + ((nvec + 3) & ~(size_t)3) * sizeof(float) + 64 * sizeof(f32x4);
}

template <int WT, int MODE>
static hipError_t launch_one(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o, const float* rays_d,
                             const float* z, const float* dirfeat, int64_t M, int S, float* sigma, float* craw, int want_rgb,
                             int nblk, int n_cu, hipStream_t stream) {
  const size_t lds = lds_bytes<WT, MODE>(net);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int ntiles = (int)((M + C::TILE - 1) / C::TILE);
  const dim3 grid((unsigned)std::min(ntiles, n_cu)), block(64 * C::WAVES);   // one persistent workgroup per CU
  hipLaunchKernelGGL((mlp_fwd_bf16_t16_kernel<WT, MODE>), grid, block, lds, stream, net, (const char*)stream_w, bvec, rays_o, rays_d,
                     z, dirfeat, M, S, sigma, craw, want_rgb, nblk, ntiles);
  return hipGetLastError();
}
template <int WT>
static hipError_t launch_mode(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o, const float* rays_d,
                              const float* z, const float* dirfeat, int64_t M, int S, float* sigma, float* craw, int want_rgb,
                              int nblk, int n_cu, hipStream_t stream) {
  switch (net.mode) {
    case 1: return launch_one<WT, 1>(net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    case 2: return launch_one<WT, 2>(net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    default: return launch_one<WT, 0>(net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
  }
}
template <int WT, int MODE>
static hipError_t prepare_one() {
  return hipFuncSetAttribute((const void*)mlp_fwd_bf16_t16_kernel<WT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace t16

// once per device, with that device current (nerfhip_create): every instantiation may use the whole 160-KiB LDS
hipError_t prepare_mlp_fwd_bf16_t16(int W, int mode) {
  switch ((W / 32) * 4 + mode) {
    case 2 * 4 + 0: return t16::prepare_one<2, 0>();
    case 2 * 4 + 1: return t16::prepare_one<2, 1>();
    case 2 * 4 + 2: return t16::prepare_one<2, 2>();
    case 4 * 4 + 0: return t16::prepare_one<4, 0>();
    case 4 * 4 + 1: return t16::prepare_one<4, 1>();
    case 4 * 4 + 2: return t16::prepare_one<4, 2>();
    case 8 * 4 + 0: return t16::prepare_one<8, 0>();
    case 8 * 4 + 1: return t16::prepare_one<8, 1>();
    case 8 * 4 + 2: return t16::prepare_one<8, 2>();
    default: return hipErrorInvalidValue;
  }
}

// stream_w: the 16x16x32-ordered full stream when want_rgb, else the density-only one (api.hip: build_fwd_streams_t16)
hipError_t launch_mlp_fwd_bf16_t16(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                                   const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                                   float* craw, int want_rgb, int nblk, int n_cu, hipStream_t stream) {
  switch (net.W / 32) {
    case 2: return t16::launch_mode<2>(net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    case 4: return t16::launch_mode<4>(net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    case 8: return t16::launch_mode<8>(net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace nh
