// mlp_fwd_f16x3.hip -- fused positional-encoding + density/colour MLP forward with SPLIT-f16 MFMA operands (gfx950),
// inference: the function of mlp_fwd.hip (reference nerf/nerfCoreBlock.scala:17-112) to fp32 accuracy at matrix-core
// rates (f16x3_common.hip.h: x = hi + lo, three v_mfma_f32_32x32x16_f16 per product into one fp32 accumulator that
// carries the factor 64 of the packed weights).
// Held to the SAME parity bars as the fp32 kernels (tests: every fp32 render test also runs with dtype f16x3).
// Positions, the encoding's sin/cos (accurate ones, mlp_common.hip.h, not the hardware approximations of the bf16 path), biases,
// the per-ray view-direction branch and all accumulators are fp32.
// TRAIN variant (forward of a training step with dtype f16x3; backward and weight gradient stay on the fp32 kernels):
// the same arithmetic, and every post-ReLU activation, the encoding, the colour hidden layer and the ReLU sign bits are
// written in fp32 into the fp32 path's activation store (common.h: ActStore, tile-major -- the accumulator layout of
// the 32x32 f16 MFMA is that of the 32x32 fp32 one, so a register goes out exactly as mlp_fwd.hip writes it).
// One wave = 32 samples x all features, one wave per SIMD (512 registers), workgroup = 4 waves = 128 samples,
// persistent (one per CU).  Stream layout = that of the bf16 kernel with every fragment replaced by its (hi, lo) pair
// and every part padded to whole 16-pair blocks (api.hip: build_bf16_streams(pad 16) + pack.hip: pack_frags_f16x3).
// -DNERFHIP_DIAG_STAMPS=1 makes a DIAGNOSTIC library (scripts/x3_stamps.sh builds its own copy under build/, never the
// shipped one): one wave records s_memtime at the part boundaries of its 4th tile and the workgroup overwrites the first
// bytes of `craw` with them -- results of that build are wrong by construction.
#ifndef NERFHIP_DIAG_STAMPS
#define NERFHIP_DIAG_STAMPS 0
#endif
#include "f16x3_common.hip.h"
#include <algorithm>
#include "launchers.h"

namespace nh {
namespace x3 {

template <int WT>
struct ActRegs { Pair v[WT][2]; };   // [k-tile][k-step]

template <int WT>
struct SelAct {   // k-steps 0 .. 2WT-1: previous layer; 2WT .. 2WT+3: encoding (skip layers)
  const ActRegs<WT>& a;
  const ActRegs<2>* e;
  template <int KS_IDX> __device__ __forceinline__ const Pair& b() const {
    if constexpr (KS_IDX < 2 * WT) return a.v[KS_IDX / 2][KS_IDX % 2];
    else return e->v[(KS_IDX - 2 * WT) / 2][KS_IDX % 2];
  }
};
struct SelEnc {
  const ActRegs<2>& e;
  template <int KS_IDX> __device__ __forceinline__ const Pair& b() const { return e.v[KS_IDX / 2][KS_IDX % 2]; }
};
template <int HT>
struct SelHid {
  const ActRegs<HT>& a;
  template <int KS_IDX> __device__ __forceinline__ const Pair& b() const { return a.v[KS_IDX / 2][KS_IDX % 2]; }
};

// In every epilogue `a` is the finished accumulator tile = 64 x (W x + b) (f16x3_common.hip.h: kWScale).

// relu on the integer bits (negative floats and -0 are negative integers): one v_max_i32 where fmaxf on a value that
// comes out of an asm-free register costs a canonicalising v_max_f32 as well
__device__ __forceinline__ float relu_bits(float x) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0)); }

// what the TRAIN epilogues need to write a tile into the activation store
struct StoreCtx {
  float* dst;          // wave-uniform: this wave's 32-sample tile block of the layer's buffer (mlp_common.hip.h: tile_block)
  uint32_t voff;       // tile_voff(lane)
  uint32_t* mask;      // this lane's word 0 of the layer's sign-bit slot (mask_ptr(.., w = 0, ..)); word w at + w * Mpad
  int64_t Mpad;
};
// a pointer that IS wave-uniform, made so for the compiler (the asm stores take their base in SGPRs)
__device__ __forceinline__ float* uniform_ptr(float* p) {
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (float*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
// sign bit of a post-ReLU value, at bit position POS of a mask word ("> 0" on the integer bits, mlp_common.hip.h)
template <int POS>
__device__ __forceinline__ uint32_t sign_bit_at(float v) { return (uint32_t)min(__builtin_bit_cast(int, v), 1) << POS; }

// Range guard (f16x3_common.hip.h: an activation beyond f16's 65504 has no high half).  An overflowed activation becomes
// hi = +inf, lo = -inf, the next layer's accumulators NaN -- and a NaN with its sign bit set is a negative integer that the
// ReLU turns into 0, so the network would quietly go on with zeros (measured: every output = its bias).  The overflow is
// therefore caught where it happens: every freshly made pair of high halves goes through ONE v_pk_max_u16 into a running
// maximum (post-ReLU halves are non-negative, so as unsigned 16-bit patterns inf = 0x7c00 and NaN > 0x7c00 are the
// largest values); signed inputs (the encoding) drop their sign bits first.  The outputs are tested as well (one integer
// test per written value).  Everything is OR-ed into one register that the kernel publishes once, at its end.
__device__ __forceinline__ uint32_t nonfinite_bits(float v) { return (__builtin_bit_cast(uint32_t, v) & 0x7f800000u) == 0x7f800000u ? 1u : 0u; }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void track_hi(uint32_t& mx, uint32_t h) {
  mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, mx), __builtin_bit_cast(u16x2, h)));
}
__device__ __forceinline__ uint32_t hi_overflowed(uint32_t mx) { return ((mx & 0xffffu) >= 0x7c00u || (mx >> 16) >= 0x7c00u) ? 1u : 0u; }

template <int WT, bool TRAIN>
struct EpiTrunk {   // relu + split -> B operand pairs of the next layer; piece P = registers 2P, 2P+1 of the tile
  ActRegs<WT>& out;
  Ring& rg;
  StoreCtx sc;
  uint32_t mw;
  uint32_t& mx;     // range guard: running maximum of the high halves made
  static constexpr int kPieces = 8;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x16& a) {
    if constexpr (!TRAIN) {
      split_scaled_into(out.v[OT_IDX][P / 4], P % 4, relu_bits(a[2 * P]), relu_bits(a[2 * P + 1]));
      track_hi(mx, out.v[OT_IDX][P / 4].hi[P % 4]);
    } else {
      const float v0 = relu_bits(a[2 * P]) * kWInv, v1 = relu_bits(a[2 * P + 1]) * kWInv;   // exact: a power of two
      store_reg<OT_IDX, 2 * P>(sc.dst, sc.voff, v0);
      store_reg<OT_IDX, 2 * P + 1>(sc.dst, sc.voff, v1);
      rg.note(2);
      if constexpr (P == 0 && (OT_IDX & 1) == 0) mw = 0u;
      mw |= sign_bit_at<16 * (OT_IDX & 1) + 2 * P>(v0) | sign_bit_at<16 * (OT_IDX & 1) + 2 * P + 1>(v1);
      if constexpr (P == kPieces - 1 && ((OT_IDX & 1) == 1 || OT_IDX == WT - 1)) {
        sc.mask[(int64_t)(OT_IDX / 2) * sc.Mpad] = mw;
        rg.note(1);
      }
      split_into(out.v[OT_IDX][P / 4], P % 4, v0, v1);
      track_hi(mx, out.v[OT_IDX][P / 4].hi[P % 4]);
    }
  }
};
struct EpiSigma {   // one tile whose row 0 is the density row (nerfCoreBlock.scala:39-40)
  float* sigma; int m; bool valid; int h; uint32_t& bad;
  static constexpr int kPieces = 1;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x16& a) {
    if (valid && h == 0) { sigma[m] = a[0] * kWInv; bad |= nonfinite_bits(a[0]); }
  }
};
template <int HT, int MODE, bool TRAIN>
struct EpiHead {    // tiles 0..HT-1: first head matrix; tile HT: density row
  ActRegs<HT>& hid; float (&c)[3];
  const float* dirfeat; int ray; int m; bool valid; float* sigma; int h;
  Ring& rg;
  StoreCtx sc;      // TRAIN, mode 1: the colour hidden layer (ActStore::hc) and its sign bits (slot D)
  uint32_t mw;
  uint32_t& bad;
  uint32_t& bad_mx;   // range guard: running maximum of the high halves made (colour hidden layer)
  static constexpr int kPieces = 8;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x16& a) {
    if constexpr (OT_IDX == HT) {
      if constexpr (P == 0)
        if (valid && h == 0) { sigma[m] = a[0] * kWInv; bad |= nonfinite_bits(a[0]); }
    } else if constexpr (MODE == 1) {
      // add().relu() (nerfCoreBlock.scala:67): + (Wd.enc(dir) + bd + bf) of the sample's ray, fp32; registers 2P, 2P+1
      // hold features 32 OT + 8 (P/2) + 2 (P%2) + 4h + {0, 1}
      const float* df = dirfeat + (int64_t)ray * (32 * HT) + 32 * OT_IDX + 8 * (P / 2) + 2 * (P % 2) + 4 * h;
      const f32x2 d = *reinterpret_cast<const f32x2*>(df);
      const float v0 = fmaxf(fmaf(a[2 * P], kWInv, d[0]), 0.f), v1 = fmaxf(fmaf(a[2 * P + 1], kWInv, d[1]), 0.f);
      if constexpr (TRAIN) {
        store_reg<OT_IDX, 2 * P>(sc.dst, sc.voff, v0);
        store_reg<OT_IDX, 2 * P + 1>(sc.dst, sc.voff, v1);
        rg.note(2);
        if constexpr (P == 0 && (OT_IDX & 1) == 0) mw = 0u;
        mw |= sign_bit_at<16 * (OT_IDX & 1) + 2 * P>(v0) | sign_bit_at<16 * (OT_IDX & 1) + 2 * P + 1>(v1);
        if constexpr (P == kPieces - 1 && ((OT_IDX & 1) == 1 || OT_IDX == HT - 1)) {
          sc.mask[(int64_t)(OT_IDX / 2) * sc.Mpad] = mw;
          rg.note(1);
        }
      }
      split_into(hid.v[OT_IDX][P / 4], P % 4, v0, v1);
      track_hi(bad_mx, hid.v[OT_IDX][P / 4].hi[P % 4]);
    } else if constexpr (MODE == 2) {
      // reshape(...,9,3).mul(SH).sum(-2) (:55): c[k] = sum_j q[3j+k] * Y_j(dir); rows 0..26 of tile 0
      const float* Y = dirfeat + (int64_t)ray * 12;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = tile_row(0, 2 * P + i, h);
        if (row < 27) {
          const int j = row / 3, k = row - 3 * j;
          const float v = (a[2 * P + i] * kWInv) * Y[j];
          c[0] += k == 0 ? v : 0.f; c[1] += k == 1 ? v : 0.f; c[2] += k == 2 ? v : 0.f;
        }
      }
    } else {
      if (h == 0) {
        if constexpr (P == 0) { c[0] = a[0] * kWInv; c[1] = a[1] * kWInv; }
        if constexpr (P == 1) c[2] = a[2] * kWInv;
      }
    }
  }
};
struct EpiColour {   // colour output layer (rows 0..2 of its single tile)
  float (&c)[3]; int h;
  static constexpr int kPieces = 2;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x16& a) {
    if (h == 0) {
      if constexpr (P == 0) { c[0] = a[0] * kWInv; c[1] = a[1] * kWInv; }
      if constexpr (P == 1) c[2] = a[2] * kWInv;
    }
  }
};

// positionCode (nerfCoreBlock.scala:127-136) of this lane's 32 of the 64 (padded) features, split into operand pairs
// (element 2i, 2i+1 of k-step s of k-tile kt = accumulator-layout register 8s + 2i, 8s + 2i + 1 of tile kt)
template <bool RED, bool TRAIN, int KT = 0, int S_ = 0, int I = 0>
__device__ __forceinline__ void encode(ActRegs<2>& enc, float px, float py, float pz, int E, int h, const StoreCtx& sc, uint32_t& mx) {
  if constexpr (KT < 2) {
    const float v0 = code_value_h<RED>(px, py, pz, 32 * KT + feature(S_, 0, 2 * I), h, E);
    const float v1 = code_value_h<RED>(px, py, pz, 32 * KT + feature(S_, 0, 2 * I + 1), h, E);
    if constexpr (TRAIN) {
      store_reg<KT, 8 * S_ + 2 * I>(sc.dst, sc.voff, v0);
      store_reg<KT, 8 * S_ + 2 * I + 1>(sc.dst, sc.voff, v1);
    }
    split_into(enc.v[KT][S_], I, v0, v1);
    if constexpr (KT == 0 && S_ == 0 && I < 2) track_hi(mx, enc.v[KT][S_].hi[I] & 0x7fff7fffu);   // the raw position (rows 0..2): the only unbounded rows
    constexpr int nxt = (KT * 2 + S_) * 4 + I + 1;
    encode<RED, TRAIN, nxt / 8, (nxt / 4) % 2, nxt % 4>(enc, px, py, pz, E, h, sc, mx);
  }
}

template <int WT, int MODE, bool TRAIN>
__global__ __launch_bounds__(64 * kWaves, 1) void mlp_fwd_f16x3_kernel(NetArgs net, const char* __restrict__ stream,
                                                                       const float* __restrict__ bvec,
                                                                       const float* __restrict__ rays_o,
                                                                       const float* __restrict__ rays_d,
                                                                       const float* __restrict__ z,
                                                                       const float* __restrict__ dirfeat, int64_t M, int S,
                                                                       float* __restrict__ sigma, float* __restrict__ craw,
                                                                       int want_rgb, int nblk, int ntiles, ActStore st,
                                                                       uint32_t* __restrict__ nonfinite_flag) {
  constexpr int R = 32 * WT;
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring (kRing x 32 KiB) | vectors
  float* vec = smem + kRing * kBlkBytes / 4;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = net.D;
  const int nvec = D * R + (HT + 1) * 32 + 32;

  Ring rg;
  rg.stream = stream;
  rg.ring = smem;
  rg.ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem + (uint32_t)lane * 16u;
  rg.wave = wave;
  rg.lane = lane;
  rg.voff = (uint32_t)(wave * (kBlkBytes / kWaves) + lane * 16);
  rg.nblk = nblk;
  for (int i = tid; i < nvec; i += 64 * kWaves) vec[i] = bvec[i];
  const uint32_t vec_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)vec + (uint32_t)h * 16u;

  __syncthreads();   // vectors visible
  rg.start();

#if NERFHIP_DIAG_STAMPS
  uint64_t stamp_t[16]; int it = 0;
#define STAMP(i) do { if (it == 3) stamp_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i)
#endif
  uint32_t bad = 0u, mx = 0u;   // range guard: this lane wrote a non-finite output / running maximum of the f16 high halves made
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    STAMP(0);
    const int m = tile * kTileSamples + wave * 32 + li;   // host guarantees M < 2^31
    const bool valid = m < (int)M;
    const int mc = valid ? m : (int)M - 1;
    const int ray = (int)((uint32_t)mc / (uint32_t)S);
    // sample position (raysO.add(raysD.mul(zVals)), nerfBlock.scala:65): separate mul and add, fp32
    const float zz = z[mc];
    const float px = __fadd_rn(rays_o[(int64_t)ray * 3 + 0], __fmul_rn(rays_d[(int64_t)ray * 3 + 0], zz));
    const float py = __fadd_rn(rays_o[(int64_t)ray * 3 + 1], __fmul_rn(rays_d[(int64_t)ray * 3 + 1], zz));
    const float pz = __fadd_rn(rays_o[(int64_t)ray * 3 + 2], __fmul_rn(rays_d[(int64_t)ray * 3 + 2], zz));
    // the encoding is computed once per sample and kept (32 registers) for the skip layers: its sin/cos are the
    // accurate ones, far too expensive to recompute
    // TRAIN: this wave's 32-sample tile of the activation store (tile index as in mlp_fwd.hip: 4 per workgroup tile)
    const int tix = tile * kWaves + wave;
    const uint32_t svoff = tile_voff(lane);
    auto store_ctx = [&](float* buf, int rows, int slot) __attribute__((always_inline)) {
      return StoreCtx{TRAIN ? uniform_ptr(tile_block(buf, tix, rows)) : nullptr, svoff, TRAIN ? mask_ptr(st.mask, slot, h, 0, st.Mpad, m) : nullptr,
                      st.Mpad};
    };
    ActRegs<2> enc;
    {
      const StoreCtx sce = store_ctx(st.enc, 64, 0);
      if (trig_reduced(px, py, pz, net.E)) encode<true, TRAIN>(enc, px, py, pz, net.E, h, sce, mx);   // wave-uniform: the usual case
      else encode<false, TRAIN>(enc, px, py, pz, net.E, h, sce, mx);
      if constexpr (TRAIN) rg.note(32);
    }

    STAMP(1);
    ActRegs<WT> act, nxt;
    {
      EpiTrunk<WT, TRAIN> epi{act, rg, store_ctx(st.act, R, 0), 0u, mx};
      run_part<4, WT, true, TRAIN>(rg, vec_addr, SelEnc{enc}, epi);   // layer 0
    }
    STAMP(2);
    auto layer = [&](int l, const ActRegs<WT>& in, ActRegs<WT>& out) __attribute__((always_inline)) {
      EpiTrunk<WT, TRAIN> epi{out, rg, store_ctx(st.act + (int64_t)l * R * st.Mpad, R, l), 0u, mx};
      const uint32_t baddr = vec_addr + (uint32_t)(l * R) * 4u;
      if ((net.skip_mask >> (l - 1)) & 1u) run_part<2 * WT + 4, WT, true, TRAIN>(rg, baddr, SelAct<WT>{in, &enc}, epi);
      else run_part<2 * WT, WT, true, TRAIN>(rg, baddr, SelAct<WT>{in, nullptr}, epi);
    };
    int l = 1;
    for (; l + 1 < D; l += 2) {
      layer(l, act, nxt);
      STAMP(2 + l);
      layer(l + 1, nxt, act);
      STAMP(3 + l);
    }
    if (l < D) {
      layer(l, act, nxt);
      act = nxt;
    }

    const uint32_t haddr = vec_addr + (uint32_t)(D * R) * 4u;
    if (!want_rgb) {
      // sigma-only stream: one tile, row 0 = density row (nerfCoreBlock.scala:39-40)
      EpiSigma epi{sigma, m, valid, h, bad};
      run_part<2 * WT, 1, true, TRAIN>(rg, haddr + (uint32_t)(HT * 32) * 4u, SelAct<WT>{act, nullptr}, epi);
      continue;
    }
    // head: HT colour tiles, then the density tile
    ActRegs<HT> hid;
    float c[3] = {0.f, 0.f, 0.f};
    EpiHead<HT, MODE, TRAIN> epih{hid, c, dirfeat, ray, m, valid, sigma, h, rg, store_ctx(st.hc, 32 * HT, D), 0u, bad, mx};
    STAMP(10);
    run_part<2 * WT, HT + 1, true, TRAIN>(rg, haddr, SelAct<WT>{act, nullptr}, epih);
    STAMP(11);
    if constexpr (MODE == 1) {
      // colour output layer: Linear(3) over the H2 hidden features, as one MFMA tile (rows 0..2)
      EpiColour epic{c, h};
      run_part<2 * HT, 1, true, TRAIN>(rg, haddr + (uint32_t)((HT + 1) * 32) * 4u, SelHid<HT>{hid}, epic);
    } else if constexpr (MODE == 2) {
      c[0] += __shfl_xor(c[0], 32); c[1] += __shfl_xor(c[1], 32); c[2] += __shfl_xor(c[2], 32);
    }
    STAMP(12);
#if NERFHIP_DIAG_STAMPS
    ++it;
#endif
    if (valid && h == 0) {
      craw[(int64_t)m * 3 + 0] = c[0]; craw[(int64_t)m * 3 + 1] = c[1]; craw[(int64_t)m * 3 + 2] = c[2];
      bad |= nonfinite_bits(c[0]) | nonfinite_bits(c[1]) | nonfinite_bits(c[2]);
    }
  }   // tile loop
  bad |= hi_overflowed(mx);
  if (bad && nonfinite_flag) atomicOr(nonfinite_flag, 1u);
#if NERFHIP_DIAG_STAMPS
  if (blockIdx.x == 5 && tid == 0) for (int i = 0; i < 13; ++i) reinterpret_cast<uint64_t*>(craw)[i] = stamp_t[i];
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

template <int WT, int MODE>
static hipError_t launch_one(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o, const float* rays_d,
                             const float* z, const float* dirfeat, int64_t M, int S, float* sigma, float* craw, int want_rgb,
                             int nblk, int n_cu, const ActStore* st, uint32_t* nonfinite_flag, hipStream_t stream) {
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  const size_t lds = (size_t)kRing * kBlkBytes + ((size_t)net.D * 32 * WT + (HT + 1) * 32 + 32) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int ntiles = (int)((M + kTileSamples - 1) / kTileSamples);
  const dim3 grid((unsigned)std::min(ntiles, n_cu)), block(64 * kWaves);   // one persistent workgroup per CU
  if (st)
    hipLaunchKernelGGL((mlp_fwd_f16x3_kernel<WT, MODE, true>), grid, block, lds, stream, net, (const char*)stream_w, bvec, rays_o,
                       rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, ntiles, *st, nonfinite_flag);
  else
    hipLaunchKernelGGL((mlp_fwd_f16x3_kernel<WT, MODE, false>), grid, block, lds, stream, net, (const char*)stream_w, bvec, rays_o,
                       rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, ntiles, ActStore{}, nonfinite_flag);
  return hipGetLastError();
}

}  // namespace x3

#define NH_X3_CASES(F)                                                                                                  \
  switch ((W / 32) * 4 + mode) {                                                                                        \
    case 2 * 4 + 0: F(2, 0); case 2 * 4 + 1: F(2, 1); case 2 * 4 + 2: F(2, 2);                                          \
    case 4 * 4 + 0: F(4, 0); case 4 * 4 + 1: F(4, 1); case 4 * 4 + 2: F(4, 2);                                          \
    case 8 * 4 + 0: F(8, 0); case 8 * 4 + 1: F(8, 1); case 8 * 4 + 2: F(8, 2);                                          \
    default: return hipErrorInvalidValue;                                                                               \
  }

// once per device, with that device current (nerfhip_create)
hipError_t prepare_mlp_fwd_f16x3(int W, int mode) {
#define NH_X3_PREP(WT_, MODE_)                                                                                                     \
  {                                                                                                                                \
    hipError_t e_ = hipFuncSetAttribute((const void*)x3::mlp_fwd_f16x3_kernel<WT_, MODE_, false>,                                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                   \
    if (e_ != hipSuccess) return e_;                                                                                               \
    return hipFuncSetAttribute((const void*)x3::mlp_fwd_f16x3_kernel<WT_, MODE_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               160 * 1024);                                                                                        \
  }
  NH_X3_CASES(NH_X3_PREP)
#undef NH_X3_PREP
}

// stream_w: the full pair stream when want_rgb, else the sigma-only one (api.hip)
// st: the fp32 activation store to fill (training forward), or null
hipError_t launch_mlp_fwd_f16x3(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                                const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                                float* craw, int want_rgb, int nblk, int n_cu, const ActStore* st, uint32_t* nonfinite_flag,
                                hipStream_t stream) {
  const int W = net.W, mode = net.mode;
#define NH_X3_LAUNCH(WT_, MODE_) \
  return x3::launch_one<WT_, MODE_>(net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, st, nonfinite_flag, stream)
  NH_X3_CASES(NH_X3_LAUNCH)
#undef NH_X3_LAUNCH
}

}  // namespace nh
