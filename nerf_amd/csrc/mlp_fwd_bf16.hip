// mlp_fwd_bf16.hip -- fused positional-encoding + density/colour MLP forward on bf16 MFMA (gfx950), inference.
//
// Same function as mlp_fwd.hip (reference nerf/nerfCoreBlock.scala:17-112) with bf16 MFMA operands and fp32
// accumulation (BASELINE config 5; parity bar: renders within 1e-2 abs of the fp32 path).  Positions, the
// encoding's sin/cos, biases, the per-ray view-direction branch and all accumulators are fp32; weights and
// the activations fed to the next layer are rounded to bf16 (RNE).  Machinery: bf16_common.hip.h.
// One wave = 32 samples x all features, workgroup = 8 waves (two per SIMD) = 256 samples, persistent (one per CU).  The
// kernel is templated on the sample tiles per wave (bf16_common.hip.h: BfCfg); NH_BF_INFER_NS below says why inference
// stays at one.
// Stream layout (api.hip: build_bf16_stream): layer 0 | layers 1..D-1 (skip layers carry 4 extra k-steps for
// the encoding) | head: HT colour-hidden tiles + one tile whose row 0 is the density row | (mode 1) colour
// output layer, 1 tile over K = 32 HT.  The sigma-only stream (coarse pass of inference) ends with the density
// tile alone.
#include "bf16_common.hip.h"
#include <algorithm>
// Sample tiles per wave of the INFERENCE kernel.  2 (four waves, one per SIMD, every LDS fragment feeding two MFMAs) was
// built and measured in round 3: it renders the same pixels (tests/test_gpu_bf16.py green) but no faster -- with the skip
// layer's encoding live the 512-register file overflows (158-227 spilled registers: 1.04-1.24 PF), and a build without
// the skip layer, where it fits, runs at the one-tile kernel's rate (1.418 against 1.440 PF; profiles/r3_bf16_ns2_*): the
// LDS read port (256 B/clk/CU for ds_read_b128, half used) was never the limit, the kernel is clock/power-limited.
#ifndef NH_BF_INFER_NS
#define NH_BF_INFER_NS 1
#endif
#include "launchers.h"

namespace nh {

template <int NS, int WT>
struct ActRegs { bf16x8 v[NS][WT][2]; };

template <int NS, int WT>
struct SelAct {   // k-steps 0 .. 2WT-1: previous layer; 2WT .. 2WT+3: encoding (skip layers)
  const ActRegs<NS, WT>& a;
  const ActRegs<NS, 2>* e;
  template <int KS_IDX> __device__ __forceinline__ const bf16x8& b(int n) const {
    if constexpr (KS_IDX < 2 * WT) return a.v[n][KS_IDX / 2][KS_IDX % 2];
    else return e->v[n][(KS_IDX - 2 * WT) / 2][KS_IDX % 2];
  }
};
template <int NS>
struct SelEnc {
  const ActRegs<NS, 2>& e;
  template <int KS_IDX> __device__ __forceinline__ const bf16x8& b(int n) const { return e.v[n][KS_IDX / 2][KS_IDX % 2]; }
};
template <int NS, int HT>
struct SelHid {
  const ActRegs<NS, HT>& a;
  template <int KS_IDX> __device__ __forceinline__ const bf16x8& b(int n) const { return a.v[n][KS_IDX / 2][KS_IDX % 2]; }
};
template <int H, int NS>
__device__ __forceinline__ void encode_half(ActRegs<NS, 2>& enc, const float (&pos)[NS][3], int E) {
#pragma unroll
  for (int n = 0; n < NS; ++n)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          enc.v[n][kt][s][e] = (__bf16)code_value_fast(pos[n][0], pos[n][1], pos[n][2], 32 * kt + bf_feature(s, H, e), E);
}

// one 16-byte store per lane: fragment (kt, s) of this wave's sample tile (BfStore layout, common.h)
__device__ __forceinline__ void store_frag(char* tile_lane, int kt, int s, const bf16x8& v) {
  // streaming data (read once, much later, by another kernel): non-temporal, so that it does not push the weight
  // streams out of the XCD's L2
  __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(tile_lane + (kt * 2 + s) * 1024));
}
template <class C, int WT, bool TRAIN>
struct EpiTrunk {   // relu + bf16 -> B operand of the next layer (TRAIN: also into the activation store)
  static constexpr int kNS = C::NS;
  ActRegs<kNS, WT>& out;
  char* dst;        // TRAIN: this wave's tile of the layer's activation buffer + lane*16
  u32x4* bits_dst;  // TRAIN: this lane's 16 B of the layer's ReLU' bits
  BfRingT<C>& rg;
  u32x4 bits;
  static constexpr int kPieces = 2 * kNS;   // (sample tile, k-step half)
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x16 (&acc)[kNS]) {
    static_assert(!TRAIN || kNS == 1, "store addressing assumes one sample tile per wave");
    out.v[P / 2][OT_IDX][P % 2] = relu_pack_half<P % 2>(acc[P / 2]);
    if constexpr (TRAIN) {
      store_frag(dst, OT_IDX, P % 2, out.v[P / 2][OT_IDX][P % 2]);
      rg.note(1);
      relu_bits_gather<OT_IDX, P % 2>(bits, out.v[P / 2][OT_IDX][P % 2]);
      if constexpr (OT_IDX == WT - 1 && P == kPieces - 1) { __builtin_nontemporal_store(bits, bits_dst); rg.note(1); }
    }
  }
};

template <int kNS>
struct EpiSigma {   // one tile whose row 0 is the density row (nerfCoreBlock.scala:39-40)
  float* sigma; const int* m; const bool* valid; int h;
  static constexpr int kPieces = 1;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x16 (&acc)[kNS]) {
#pragma unroll
    for (int n = 0; n < kNS; ++n)
      if (valid[n] && h == 0) sigma[m[n]] = acc[n][0];
  }
};
template <int kNS, int HT, int MODE, bool TRAIN>
struct EpiHead {    // tiles 0..HT-1: first head matrix; tile HT: density row
  char* hc_dst;     // TRAIN, mode 1: this wave's tile of the hc buffer + lane*16
  ActRegs<kNS, HT>& hid; float (&c)[kNS][3];
  const float* dirfeat; const int* ray; const int* m; const bool* valid; float* sigma; int h;
  static constexpr int kPieces = 1;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x16 (&acc)[kNS]) {
    if constexpr (OT_IDX == HT) {
#pragma unroll
      for (int n = 0; n < kNS; ++n)
        if (valid[n] && h == 0) sigma[m[n]] = acc[n][0];
    } else if constexpr (MODE == 1) {
      // add().relu() (nerfCoreBlock.scala:67): + (Wd.enc(dir) + bd + bf) of the sample's ray, fp32
#pragma unroll
      for (int n = 0; n < kNS; ++n) {
        const float* df = dirfeat + (int64_t)ray[n] * (32 * HT) + 32 * OT_IDX + 4 * h;
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(df + 8 * q);
          v[4 * q] = acc[n][4 * q] + d[0]; v[4 * q + 1] = acc[n][4 * q + 1] + d[1];
          v[4 * q + 2] = acc[n][4 * q + 2] + d[2]; v[4 * q + 3] = acc[n][4 * q + 3] + d[3];
        }
        relu_pack(v, hid.v[n][OT_IDX][0], hid.v[n][OT_IDX][1]);
        if constexpr (TRAIN) { store_frag(hc_dst, OT_IDX, 0, hid.v[n][OT_IDX][0]); store_frag(hc_dst, OT_IDX, 1, hid.v[n][OT_IDX][1]); }
      }
    } else if constexpr (MODE == 2) {
      // reshape(...,9,3).mul(SH).sum(-2) (:55): c[k] = sum_j q[3j+k] * Y_j(dir); rows 0..26 of tile 0
#pragma unroll
      for (int n = 0; n < kNS; ++n) {
        const float* Y = dirfeat + (int64_t)ray[n] * 12;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = tile_row(0, r, h);
          if (row < 27) {
            const int j = row / 3, k = row - 3 * j;
            const float v = acc[n][r] * Y[j];
            c[n][0] += k == 0 ? v : 0.f; c[n][1] += k == 1 ? v : 0.f; c[n][2] += k == 2 ? v : 0.f;
          }
        }
      }
    } else {
#pragma unroll
      for (int n = 0; n < kNS; ++n)
        if (h == 0) { c[n][0] = acc[n][0]; c[n][1] = acc[n][1]; c[n][2] = acc[n][2]; }
    }
  }
};
template <int kNS>
struct EpiColour {   // colour output layer (rows 0..2 of its single tile)
  float (&c)[kNS][3]; int h;
  static constexpr int kPieces = 1;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x16 (&acc)[kNS]) {
#pragma unroll
    for (int n = 0; n < kNS; ++n)
      if (h == 0) { c[n][0] = acc[n][0]; c[n][1] = acc[n][1]; c[n][2] = acc[n][2]; }
  }
};

template <int WT, int MODE, bool TRAIN, int NS>
__global__ __launch_bounds__(64 * (NS == 1 ? 8 : 4)) void mlp_fwd_bf16_kernel(NetArgs net, const char* __restrict__ stream,
                                                              const float* __restrict__ bvec,
                                                              const float* __restrict__ rays_o,
                                                              const float* __restrict__ rays_d,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ dirfeat, int64_t M, int S,
                                                              float* __restrict__ sigma, float* __restrict__ craw,
                                                              int want_rgb, int nblk, int ntiles, BfStore st) {
  using C = BfCfg<NS, (NS == 1 ? 8 : 4)>;
  constexpr int kNS = NS, kBfWaves = C::WAVES, kBfTile = C::TILE;
  static_assert(!TRAIN || NS == 1, "the activation store is written one sample tile per wave");
  constexpr int R = 32 * WT;
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring (kRing x 32 KiB) | vectors
  float* vec = smem + kRing * kBlkBytes / 4;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = net.D;
  const int nvec = D * R + (HT + 1) * 32 + 32;

  BfRingT<C> rg;
  rg.stream = stream;
  rg.ring = smem;
  rg.ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem + (uint32_t)lane * 16u;
  rg.wave = wave;
  rg.lane = lane;
  rg.nblk = nblk;
  for (int i = tid; i < nvec; i += 64 * kBfWaves) vec[i] = bvec[i];
  const uint32_t vec_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)vec + (uint32_t)h * 16u;

  __syncthreads();   // vectors visible
  rg.start();

  // Persistent: the workgroup walks sample tiles blockIdx.x, +gridDim.x, ...; the weight ring never stops (the
  // stream is consumed cyclically), so only the first tile pays its start-up latency.
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  int m[kNS], ray[kNS];   // host guarantees M < 2^31
  bool valid[kNS];
  float pos[kNS][3];
#pragma unroll
  for (int n = 0; n < kNS; ++n) {
    m[n] = tile * kBfTile + wave * (32 * kNS) + n * 32 + li;
    valid[n] = m[n] < (int)M;
    const int mc = valid[n] ? m[n] : (int)M - 1;
    ray[n] = (int)((uint32_t)mc / (uint32_t)S);
    // sample position (raysO.add(raysD.mul(zVals)), nerfBlock.scala:65): separate mul and add, fp32
    const float zz = z[mc];
#pragma unroll
    for (int k = 0; k < 3; ++k) pos[n][k] = __fadd_rn(rays_o[(int64_t)ray[n] * 3 + k], __fmul_rn(rays_d[(int64_t)ray[n] * 3 + k], zz));
  }
  // The encoding (2 k-tiles of B operand) is recomputed where it is needed (layer 0, skip layers) instead of
  // being kept live through the whole trunk: 3 VALU per value with the hardware sin/cos, and 32 registers less.
  auto encode = [&](ActRegs<kNS, 2>& enc) {
    // the feature a register element holds depends on the lane half only through "+4h": one pass per half with
    // every index a compile-time constant (3 VALU per value), instead of per-lane index arithmetic
    if (h == 0) encode_half<0, kNS>(enc, pos, net.E);
    else encode_half<1, kNS>(enc, pos, net.E);
  };
  // TRAIN: this wave's sample tile in the activation store
  const int64_t tix = (int64_t)tile * kBfWaves + wave;
  auto tile_lane = [&](uint16_t* buf, int kt_count) {
    return reinterpret_cast<char*>(buf) + tix * (kt_count * 2048) + lane * 16;
  };
  auto bits_lane = [&](int l) {   // this lane's 16 B of layer l's ReLU' bits (BfStore::relu)
    return reinterpret_cast<u32x4*>(st.relu + (int64_t)l * st.Mpad * 8) + tix * 64 + lane;
  };
  ActRegs<kNS, WT> act, nxt;
  {
    ActRegs<kNS, 2> enc;
    encode(enc);
    if constexpr (TRAIN) {
      char* e = tile_lane(st.enc, 2);
      store_frag(e, 0, 0, enc.v[0][0][0]); store_frag(e, 0, 1, enc.v[0][0][1]);
      store_frag(e, 1, 0, enc.v[0][1][0]); store_frag(e, 1, 1, enc.v[0][1][1]);
    }
    EpiTrunk<C, WT, TRAIN> epi{act, TRAIN ? tile_lane(st.act, WT) : nullptr, TRAIN ? bits_lane(0) : nullptr, rg, {0u, 0u, 0u, 0u}};
    run_part<4, WT>(rg, vec_addr, SelEnc<kNS>{enc}, epi);   // layer 0
  }
  // layers 1..D-1, two per trip (act -> nxt -> act) so the activations never have to be copied
  auto layer = [&](int l, const ActRegs<kNS, WT>& in, ActRegs<kNS, WT>& out) {
    EpiTrunk<C, WT, TRAIN> epi{out, TRAIN ? tile_lane(st.act + (int64_t)l * R * st.Mpad, WT) : nullptr, TRAIN ? bits_lane(l) : nullptr, rg, {0u, 0u, 0u, 0u}};
    const uint32_t baddr = vec_addr + (uint32_t)(l * R) * 4u;
    if ((net.skip_mask >> (l - 1)) & 1u) {
      // recompute the encoding HERE: made opaque so that it is not hoisted out of the layer loop (it would then be
      // live -- and spilled -- through the whole trunk)
#pragma unroll
      for (int n = 0; n < kNS; ++n) asm volatile("" : "+v"(pos[n][0]), "+v"(pos[n][1]), "+v"(pos[n][2]));
      ActRegs<kNS, 2> enc;
      encode(enc);
      run_part<2 * WT + 4, WT>(rg, baddr, SelAct<kNS, WT>{in, &enc}, epi);
    } else {
      run_part<2 * WT, WT>(rg, baddr, SelAct<kNS, WT>{in, nullptr}, epi);
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
    EpiSigma<kNS> epi{sigma, m, valid, h};
    run_part<2 * WT, 1>(rg, haddr + (uint32_t)(HT * 32) * 4u, SelAct<kNS, WT>{act, nullptr}, epi);
    continue;
  }

  // head: HT colour tiles, then the density tile
  ActRegs<kNS, HT> hid;
  float c[kNS][3];
#pragma unroll
  for (int n = 0; n < kNS; ++n) c[n][0] = c[n][1] = c[n][2] = 0.f;
  EpiHead<kNS, HT, MODE, TRAIN> epih{(TRAIN && MODE == 1) ? tile_lane(st.hc, HT) : nullptr, hid, c, dirfeat, ray, m, valid, sigma, h};
  run_part<2 * WT, HT + 1>(rg, haddr, SelAct<kNS, WT>{act, nullptr}, epih);

  if constexpr (MODE == 1) {
    // colour output layer: Linear(3) over the H2 hidden features, as one MFMA tile (rows 0..2)
    EpiColour<kNS> epic{c, h};
    run_part<2 * HT, 1>(rg, haddr + (uint32_t)((HT + 1) * 32) * 4u, SelHid<kNS, HT>{hid}, epic);
  } else if constexpr (MODE == 2) {
#pragma unroll
    for (int n = 0; n < kNS; ++n) {
      c[n][0] += __shfl_xor(c[n][0], 32); c[n][1] += __shfl_xor(c[n][1], 32); c[n][2] += __shfl_xor(c[n][2], 32);
    }
  }
#pragma unroll
  for (int n = 0; n < kNS; ++n)
    if (valid[n] && h == 0) { craw[(int64_t)m[n] * 3 + 0] = c[n][0]; craw[(int64_t)m[n] * 3 + 1] = c[n][1]; craw[(int64_t)m[n] * 3 + 2] = c[n][2]; }
  }   // tile loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

template <int WT, int MODE, bool TRAIN>
static hipError_t launch_one(const BfStore& st, const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                             const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                             float* craw, int want_rgb, int nblk, int n_cu, hipStream_t stream) {
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  const size_t lds = (size_t)kRing * kBlkBytes + ((size_t)net.D * 32 * WT + (HT + 1) * 32 + 32) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  constexpr int NS = TRAIN ? 1 : NH_BF_INFER_NS;   // inference: sample tiles per wave (2: four waves, BfCfg<2, 4>)
  using C = BfCfg<NS, (NS == 1 ? 8 : 4)>;
  const int ntiles = (int)((M + C::TILE - 1) / C::TILE);
  const dim3 grid((unsigned)std::min(ntiles, n_cu)), block(64 * C::WAVES);   // one persistent workgroup per CU
  hipLaunchKernelGGL((mlp_fwd_bf16_kernel<WT, MODE, TRAIN, NS>), grid, block, lds, stream, net, (const char*)stream_w, bvec, rays_o,
                     rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, ntiles, st);
  return hipGetLastError();
}

template <int WT>
static hipError_t launch_mode(const BfStore* st, const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                              const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                              float* craw, int want_rgb, int nblk, int n_cu, hipStream_t stream) {
  switch (net.mode) {
    case 1: return st ? launch_one<WT, 1, true>(*st, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream) : launch_one<WT, 1, false>(BfStore{}, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    case 2: return st ? launch_one<WT, 2, true>(*st, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream) : launch_one<WT, 2, false>(BfStore{}, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    default: return st ? launch_one<WT, 0, true>(*st, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream) : launch_one<WT, 0, false>(BfStore{}, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
  }
}

// once per device, with that device current (nerfhip_create): every instantiation may use the whole 160-KiB LDS
template <int WT, int MODE>
static hipError_t prepare_fwd_one() {
  hipError_t e = hipFuncSetAttribute((const void*)mlp_fwd_bf16_kernel<WT, MODE, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)mlp_fwd_bf16_kernel<WT, MODE, false, NH_BF_INFER_NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
hipError_t prepare_mlp_fwd_bf16(int W, int mode) {
  switch ((W / 32) * 4 + mode) {
    case 2 * 4 + 0: return prepare_fwd_one<2, 0>();
    case 2 * 4 + 1: return prepare_fwd_one<2, 1>();
    case 2 * 4 + 2: return prepare_fwd_one<2, 2>();
    case 4 * 4 + 0: return prepare_fwd_one<4, 0>();
    case 4 * 4 + 1: return prepare_fwd_one<4, 1>();
    case 4 * 4 + 2: return prepare_fwd_one<4, 2>();
    case 8 * 4 + 0: return prepare_fwd_one<8, 0>();
    case 8 * 4 + 1: return prepare_fwd_one<8, 1>();
    case 8 * 4 + 2: return prepare_fwd_one<8, 2>();
    default: return hipErrorInvalidValue;
  }
}

// stream_w: the full stream when want_rgb, else the sigma-only stream (api.hip); st != nullptr: training forward
// (activations also go to the bf16 store)
hipError_t launch_mlp_fwd_bf16(const BfStore* st, const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                               const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                               float* craw, int want_rgb, int nblk, int n_cu, hipStream_t stream) {
  switch (net.W / 32) {
    case 2: return launch_mode<2>(st, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    case 4: return launch_mode<4>(st, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    case 8: return launch_mode<8>(st, net, stream_w, bvec, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, nblk, n_cu, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace nh
