// mlp_bwd_bf16.hip -- backward-data pass of the density/colour MLP on bf16 MFMA (gfx950).
//
// Reverse mode of nerfCoreBlock.forwardInternal (reference nerf/nerfCoreBlock.scala:17-112; the reference gets it
// from libtorch autograd, runNerf.scala:96-105) with bf16 MFMA operands and fp32 accumulation, same machinery as
// the bf16 forward (bf16_common.hip.h): the gradient tile in the accumulators becomes -- masked by ReLU', rounded
// to bf16 -- the B operand of  dX^T = W_l^T . dY_l^T  and is stored (operand layout, common.h: BfStore) for the
// bf16 weight-gradient kernel.  ReLU' comes from the stored bf16 activations (a value is > 0 iff its bf16 is
// non-zero), applied to packed pairs: mask = 0 - min(act, 1) per 16-bit half, three VALU per two values.
// Backward stream (api.hip: build_bf16_bwd_stream), all matrices transposed:
//   (mode 1) colour output layer Wc^T: HT tiles x 1 k-step | head: WT tiles x (2 HT k-steps of Wf^T + 1 k-step
//   whose k = 0 is the density row) | layers D-1 .. 1: WT tiles x 2 WT k-steps of W_l^T (first W input columns).
#include <algorithm>
#include "bf16_common.hip.h"
#include "launchers.h"

namespace nh {

template <int WT>
struct GradRegs { bf16x8 v[kNS][WT][2]; };

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// registers 8 s .. 8 s + 7 of an accumulator tile, rounded to bf16 and zeroed where the stored activation is zero
template <int S_>
__device__ __forceinline__ bf16x8 mask_pack_half(const f32x16& a, const bf16x8& act) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 v = {a[8 * S_ + 2 * i], a[8 * S_ + 2 * i + 1]};
    r[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  }
  const u16x8 one = {1, 1, 1, 1, 1, 1, 1, 1};
  const i16x8 m = -__builtin_bit_cast(i16x8, __builtin_elementwise_min(__builtin_bit_cast(u16x8, act), one));
  return __builtin_bit_cast(bf16x8, __builtin_bit_cast(i16x8, r) & m);
}

__device__ __forceinline__ bf16x8 load_frag(const char* tile_lane, int kt, int s) {
  return *reinterpret_cast<const bf16x8*>(tile_lane + (kt * 2 + s) * 1024);
}
__device__ __forceinline__ void store_frag_b(char* tile_lane, int kt, int s, const bf16x8& v) {
  // streaming data (read once, much later, by another kernel): non-temporal, so that it does not push the weight
  // streams out of the XCD's L2
  __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(tile_lane + (kt * 2 + s) * 1024));
}

template <int WT>
struct SelGrad {
  const GradRegs<WT>& g;
  template <int KS_IDX> __device__ __forceinline__ const bf16x8& b(int n) const { return g.v[n][KS_IDX / 2][KS_IDX % 2]; }
};
template <int HT>
struct SelHead {    // k-steps 0..2HT-1: d(head pre-activation); 2HT: the density k-step (k = 0 is d sigma)
  const GradRegs<HT>& dh;
  const bf16x8& dsig;
  template <int KS_IDX> __device__ __forceinline__ const bf16x8& b(int n) const {
    if constexpr (KS_IDX < 2 * HT) return dh.v[n][KS_IDX / 2][KS_IDX % 2];
    else return dsig;
  }
};
struct SelOne {
  const bf16x8& x;
  template <int KS_IDX> __device__ __forceinline__ const bf16x8& b(int n) const { return x; }
};

// epilogue of a part whose output tiles are gradients w.r.t. the pre-activations of a ReLU layer with NT tiles:
// mask by the stored activations (fetched one tile ahead), keep as next B operand, store for the weight gradient
template <int NT>
struct EpiGrad {
  GradRegs<NT>& out;
  const char* act;   // stored post-ReLU activations of that layer, this wave's tile + lane*16
  char* dst;         // gradient store, this wave's tile + lane*16
  BfRing& rg;
  bf16x8 mk[2][2];
  static constexpr int kPieces = 2;
  __device__ __forceinline__ void prefetch(int ot) { mk[ot & 1][0] = load_frag(act, ot, 0); mk[ot & 1][1] = load_frag(act, ot, 1); rg.note(2); }
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x16 (&acc)[kNS]) {
    if constexpr (P == 0 && OT_IDX + 1 < NT) prefetch(OT_IDX + 1);
    out.v[0][OT_IDX][P] = mask_pack_half<P>(acc[0], mk[OT_IDX & 1][P]);
    store_frag_b(dst, OT_IDX, P, out.v[0][OT_IDX][P]);
    rg.note(1);
  }
};

// the same for a trunk layer, masked by the ReLU' bits the forward kernel stored (16 B per lane per layer, fetched
// a whole layer ahead -- the stored activations themselves are only read by the weight-gradient kernel)
template <int NT>
struct EpiGradBits {
  GradRegs<NT>& out;
  char* dst;          // gradient store, this wave's tile + lane*16
  BfRing& rg;
  const u32x4& bits;
  static constexpr int kPieces = 2;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(f32x16 (&acc)[kNS]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 v = {acc[0][8 * P + 2 * i], acc[0][8 * P + 2 * i + 1]};
      r[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    const u32x4 m = relu_bits_expand<OT_IDX, P>(bits);
    out.v[0][OT_IDX][P] = __builtin_bit_cast(bf16x8, r & m);
    store_frag_b(dst, OT_IDX, P, out.v[0][OT_IDX][P]);
    rg.note(1);
  }
};

template <int WT, int MODE>
__global__ __launch_bounds__(64 * kBfWaves) void mlp_bwd_bf16_kernel(NetArgs net, const char* __restrict__ stream,
                                                                      const float* __restrict__ d_sigma,
                                                                      const float* __restrict__ d_craw,
                                                                      const float* __restrict__ dirfeat, int64_t M, int S,
                                                                      int nblk, int ntiles, BfStore st) {
  static_assert(kNS == 1, "one sample tile per wave");
  constexpr int R = 32 * WT;
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring only

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = net.D;

  BfRing rg;
  rg.stream = stream;
  rg.ring = smem;
  rg.ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem + (uint32_t)lane * 16u;
  rg.wave = wave;
  rg.lane = lane;
  rg.nblk = nblk;
  rg.start();

  auto bits_lane = [&](int l, int64_t tix_) {   // a lane's 16 B of layer l's ReLU' bits (BfStore::relu)
    return reinterpret_cast<const u32x4*>(st.relu + (int64_t)l * st.Mpad * 8) + tix_ * 64 + lane;
  };
  u32x4 bitsA = {0u, 0u, 0u, 0u}, bitsB = bitsA, bitsN = bitsA;
  if ((int)blockIdx.x < ntiles) bitsA = __builtin_nontemporal_load(bits_lane(D - 1, (int64_t)blockIdx.x * kBfWaves + wave));
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m = tile * kBfTile + wave * 32 + li;
    const bool valid = m < (int)M;
    const int mc = valid ? m : (int)M - 1;
    const int ray = (int)((uint32_t)mc / (uint32_t)S);
    const float dsg = valid ? d_sigma[m] : 0.f;
    float dc[3] = {0.f, 0.f, 0.f};
    if (valid) { dc[0] = d_craw[(int64_t)m * 3]; dc[1] = d_craw[(int64_t)m * 3 + 1]; dc[2] = d_craw[(int64_t)m * 3 + 2]; }

    const int64_t tix = (int64_t)tile * kBfWaves + wave;
    auto tile_lane = [&](uint16_t* buf, int kt_count) {
      return reinterpret_cast<char*>(buf) + tix * (kt_count * 2048) + lane * 16;
    };
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // feature 0 (= element 0 of lane half 0 of k-step 0) = d sigma; features 1..3 = d craw: the density k-step of
    // the head part, and tile HT of the dh store (rows of the density / colour-output weight gradients)
    bf16x8 dsig = zero8, dcol = zero8;
    if (h == 0) {
      dsig[0] = (__bf16)dsg;
      dcol[0] = (__bf16)dc[0]; dcol[1] = (__bf16)dc[1]; dcol[2] = (__bf16)dc[2];
    }
    char* dh_dst = tile_lane(st.dh, HT + 1);
    {
      bf16x8 t = dsig;
      if (h == 0) { t[1] = dcol[0]; t[2] = dcol[1]; t[3] = dcol[2]; }
      store_frag_b(dh_dst, HT, 0, t);
      store_frag_b(dh_dst, HT, 1, zero8);
    }

    // ---- d(head pre-activation)
    GradRegs<HT> dh;
    if constexpr (MODE == 1) {
      EpiGrad<HT> epi{dh, tile_lane(st.hc, HT), dh_dst, rg, {}};
      epi.prefetch(0);
      run_part<1, HT, false>(rg, 0u, SelOne{dcol}, epi);   // Wc^T . d craw, masked by the colour hidden layer's ReLU
    } else {
      // mode 2: d q[3j+k] = d c[k] * Y_j(dir) (rows 0..26); mode 0: rows 0..2 = d c
      const float* Y = dirfeat + (int64_t)ray * 12;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 t = zero8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = bf_feature(s, 0, e) + 4 * h;
          float v = 0.f;
          if (MODE == 2) { if (row < 27) v = dc[row % 3] * Y[row / 3]; }
          else if (row < 3) v = dc[row];
          t[e] = (__bf16)v;
        }
        dh.v[0][0][s] = t;
        store_frag_b(dh_dst, 0, s, t);
      }
    }

    // ---- head backward: d feat = Wf^T . dh + w_sigma . d sigma, masked by the last trunk layer's ReLU
    // ReLU' bits: the part that produces dY_k uses bits k and fetches bits k-1 (the last part: the next sample
    // tile's bits D-1), alternating between two register sets
    const int64_t tnext = (int64_t)(tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile) * kBfWaves + wave;
    GradRegs<WT> g, gn;
    {
      if (D >= 2) { bitsB = __builtin_nontemporal_load(bits_lane(D - 2, tix)); rg.note(1); }
      else { bitsN = __builtin_nontemporal_load(bits_lane(D - 1, tnext)); rg.note(1); }
      EpiGradBits<WT> epi{g, tile_lane(st.dy + (int64_t)(D - 1) * R * st.Mpad, WT), rg, bitsA};
      run_part<2 * HT + 1, WT, false>(rg, 0u, SelHead<HT>{dh, dsig}, epi);
    }
    // ---- trunk, last layer first: dY_{l-1} = ReLU'_{l-1} . (W_l^T . dY_l); two layers per trip (g -> gn -> g)
    auto layer = [&](int l, const GradRegs<WT>& in, GradRegs<WT>& out, const u32x4& use, u32x4& fetch) {
      if (l >= 2) fetch = __builtin_nontemporal_load(bits_lane(l - 2, tix));
      else bitsN = __builtin_nontemporal_load(bits_lane(D - 1, tnext));
      rg.note(1);
      EpiGradBits<WT> epi{out, tile_lane(st.dy + (int64_t)(l - 1) * R * st.Mpad, WT), rg, use};
      run_part<2 * WT, WT, false>(rg, 0u, SelGrad<WT>{in}, epi);
    };
    int l = D - 1;
    for (; l >= 2; l -= 2) {
      layer(l, g, gn, bitsB, bitsA);
      layer(l - 1, gn, g, bitsA, bitsB);
    }
    if (l >= 1) layer(l, g, gn, bitsB, bitsA);
    bitsA = bitsN;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

template <int WT, int MODE>
static hipError_t launch_bwd_one(const BfStore& st, const NetArgs& net, const void* stream_w, const float* d_sigma,
                                 const float* d_craw, const float* dirfeat, int64_t M, int S, int nblk, int n_cu, hipStream_t stream) {
  const size_t lds = (size_t)kRing * kBlkBytes;
  const int ntiles = (int)((M + kBfTile - 1) / kBfTile);
  const dim3 grid((unsigned)std::min(ntiles, n_cu)), block(64 * kBfWaves);
  hipLaunchKernelGGL((mlp_bwd_bf16_kernel<WT, MODE>), grid, block, lds, stream, net, (const char*)stream_w, d_sigma, d_craw,
                     dirfeat, M, S, nblk, ntiles, st);
  return hipGetLastError();
}

template <int WT>
static hipError_t launch_bwd_mode(const BfStore& st, const NetArgs& net, const void* stream_w, const float* d_sigma,
                                  const float* d_craw, const float* dirfeat, int64_t M, int S, int nblk, int n_cu, hipStream_t stream) {
  switch (net.mode) {
    case 1: return launch_bwd_one<WT, 1>(st, net, stream_w, d_sigma, d_craw, dirfeat, M, S, nblk, n_cu, stream);
    case 2: return launch_bwd_one<WT, 2>(st, net, stream_w, d_sigma, d_craw, dirfeat, M, S, nblk, n_cu, stream);
    default: return launch_bwd_one<WT, 0>(st, net, stream_w, d_sigma, d_craw, dirfeat, M, S, nblk, n_cu, stream);
  }
}

// once per device, with that device current (nerfhip_create)
hipError_t prepare_mlp_bwd_bf16(int W, int mode) {
  auto set = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
  switch ((W / 32) * 4 + mode) {
    case 2 * 4 + 0: return set((const void*)mlp_bwd_bf16_kernel<2, 0>);
    case 2 * 4 + 1: return set((const void*)mlp_bwd_bf16_kernel<2, 1>);
    case 2 * 4 + 2: return set((const void*)mlp_bwd_bf16_kernel<2, 2>);
    case 4 * 4 + 0: return set((const void*)mlp_bwd_bf16_kernel<4, 0>);
    case 4 * 4 + 1: return set((const void*)mlp_bwd_bf16_kernel<4, 1>);
    case 4 * 4 + 2: return set((const void*)mlp_bwd_bf16_kernel<4, 2>);
    case 8 * 4 + 0: return set((const void*)mlp_bwd_bf16_kernel<8, 0>);
    case 8 * 4 + 1: return set((const void*)mlp_bwd_bf16_kernel<8, 1>);
    case 8 * 4 + 2: return set((const void*)mlp_bwd_bf16_kernel<8, 2>);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_mlp_bwd_bf16(const BfStore& st, const NetArgs& net, const void* stream_w, const float* d_sigma,
                               const float* d_craw, const float* dirfeat, int64_t M, int S, int nblk, int n_cu, hipStream_t stream) {
  switch (net.W / 32) {
    case 2: return launch_bwd_mode<2>(st, net, stream_w, d_sigma, d_craw, dirfeat, M, S, nblk, n_cu, stream);
    case 4: return launch_bwd_mode<4>(st, net, stream_w, d_sigma, d_craw, dirfeat, M, S, nblk, n_cu, stream);
    case 8: return launch_bwd_mode<8>(st, net, stream_w, d_sigma, d_craw, dirfeat, M, S, nblk, n_cu, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace nh
