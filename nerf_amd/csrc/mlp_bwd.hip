// mlp_bwd.hip -- backward-data pass of the density/colour MLP on fp32 MFMA (gfx950).
//
// Reverse mode of nerfCoreBlock.forwardInternal (reference nerf/nerfCoreBlock.scala:17-112; the reference
// gets it from libtorch autograd via DJL's GradientCollector, runNerf.scala:96-105).  Same transposed
// formulation as the forward kernel (mlp_common.hip.h): the gradient w.r.t. a layer's pre-activations,
// dY_l^T (W x 32 samples per wave), lives in accumulator registers and is directly the B operand of
//     dX^T = W_l^T . dY_l^T
// with W_l^T read straight from the packed backward stream in L2 through the continuous fragment ring of
// mlp_common.hip.h.  The ReLU masks are the sign bitmasks written by the training forward pass; every dY_l^T
// (and the head gradients) is stored feature-major for the weight-gradient GEMM (wgrad.hip), two rows per
// tile step between the fragment loads.  Positions carry no gradient, so layer 0 and
// the encoding half of a skip input produce no dX.
#include "mlp_common.hip.h"

namespace nh {

// side operations of one trunk layer: the ReLU mask words of the NEXT layer to be processed (loaded at step 0,
// landed long before the layer ends) and the stores of this layer's dY^T
template <int WT>
struct BwdSide {
  static constexpr int NW = (WT + 1) / 2;
  using St = StoreSide<WT, WT * WT>;
  St st;
  const uint32_t* mask_next;   // slot base of layer l-1
  uint32_t* mnext;
  int64_t Mp;
  uint32_t mvoff;
  static constexpr int ops(int T) { return St::ops(T) + (T == 0 ? NW : 0); }
  template <int T> __device__ __forceinline__ void issue() const {
    if constexpr (T == 0) {
#pragma unroll
      for (int w = 0; w < NW; ++w) gload4u(mnext[w], mask_next + (int64_t)w * Mp, mvoff);
    }
    st.template issue<T>();
  }
  // The mask words were requested at step 0; step kNB-1's counted wait retires them (in order).  Until then they
  // are named by every wait so the compiler keeps them where the loads will land (it would otherwise park the
  // "values" in AGPRs right behind the load asm, i.e. copy registers whose data has not arrived yet).
  template <int T> __device__ __forceinline__ void after_wait() const {
    if constexpr (T < kNB) asm volatile("" : "+v"(mnext[0]), "+v"(mnext[1]), "+v"(mnext[2]), "+v"(mnext[3]));
  }
};

template <int WT>
__global__ __launch_bounds__(256, 1) void mlp_bwd_f32_kernel(NetArgs net, const float* __restrict__ d_sigma,
                                                             const float* __restrict__ d_craw,
                                                             const float* __restrict__ dirfeat, int64_t M, int S,
                                                             int has_rgb, ActStore st) {
  constexpr int R = 32 * WT;
  constexpr int HTM = WT >= 2 ? WT / 2 : 1;
  constexpr bool kFast = WT >= 4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int64_t m = (int64_t)blockIdx.x * kTile + wave * 32 + li;
  const bool valid = m < M;
  const int64_t mc = valid ? m : M - 1;
  const int64_t ray = mc / S;
  const float* vp = net.vecpack;
  const int W = net.W, D = net.D, HT = net.HT, Rh = 32 * net.HT;
  const int64_t Mp = st.Mpad;
  const uint32_t voff = tile_voff(lane);                                        // activation-store lane offset
  const uint32_t mvoff = mask_voff(h, Mp, m);
  const int tix = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave);   // this wave's 32-sample tile
  float* const dhb = tile_block(st.dh, tix, Rh + 32);

  // backward stream: HT head chunks (Wf^T), then WT chunks per trunk layer D-1 .. 1, all with R rows
  const float* wp = (const float*)net.wpack_bwd;
  const uint32_t fvoff = frag_voff(lane);
  f32x4 fr[kNB][4];

  const float dsig = valid ? d_sigma[m] : 0.f;
  float dc0 = 0.f, dc1 = 0.f, dc2 = 0.f;
  if (valid && has_rgb) { dc0 = d_craw[m * 3]; dc1 = d_craw[m * 3 + 1]; dc2 = d_craw[m * 3 + 2]; }

  // ---- head backward: dh = d(loss)/d(head pre-activation), in operand layout
  f32x16 dh[HTM];
#pragma unroll
  for (int t = 0; t < HTM; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[t][r] = 0.f;
  if (has_rgb) {
    if (net.mode == 1) {
      const float* wc = vp + (D + 2) * W;
      uint32_t hmask[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) hmask[w] = (w < (HTM + 1) / 2) ? *mask_ptr(st.mask, D, h, w, Mp, m) : 0u;
#pragma unroll
      for (int t = 0; t < HTM; ++t) {
        if (t < HT) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 32 * t + 8 * q + 4 * h;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wc + row);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wc + Rh + row);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wc + 2 * Rh + row);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = fmaf(w2[j], dc2, fmaf(w1[j], dc1, w0[j] * dc0));
              dh[t][4 * q + j] = ((hmask[t >> 1] >> (16 * (t & 1) + 4 * q + j)) & 1u) ? g : 0.f;
            }
          }
        }
      }
    } else if (net.mode == 2) {
      const float* Y = dirfeat + ray * 12;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tile_row(0, r, h);
        if (row < 27) {
          const int j = row / 3, k = row - 3 * j;
          dh[0][r] = (k == 0 ? dc0 : (k == 1 ? dc1 : dc2)) * Y[j];
        }
      }
    } else {
      if (h == 0) { dh[0][0] = dc0; dh[0][1] = dc1; dh[0][2] = dc2; }
    }
  }
  // head gradients go out feature-major: rows < Rh, then d sigma, then d craw^T
  if (h == 0) {
    dhb[(Rh + 0) * 32 + li] = dsig;
    dhb[(Rh + 1) * 32 + li] = dc0;
    dhb[(Rh + 2) * 32 + li] = dc1;
    dhb[(Rh + 3) * 32 + li] = dc2;
  }
  auto store_dh = [&]() { store_tiles<HTM>(dhb, dh, HT, voff); };

  // ---- d feat = w_sigma * d sigma + Wf^T . dh
  f32x16 acc[WT], g[WT];
  {
    const float* ws = vp + D * W;
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + 32 * t + 8 * q + 4 * h);
        acc[t][4 * q + 0] = wv[0] * dsig; acc[t][4 * q + 1] = wv[1] * dsig;
        acc[t][4 * q + 2] = wv[2] * dsig; acc[t][4 * q + 3] = wv[3] * dsig;
      }
  }
  if constexpr (kFast) {
    // The fragment ring is primed directly in front of its first part (asm loads are only in flight inside a
    // part, mlp_common.hip.h) and then runs through the whole stream.  The head chunks are walked even without a
    // colour gradient (dh = 0 then), which keeps the control flow free of un-primed paths.
    prime_frags(fr, wp, fvoff);
    mm_layer_g<WT, R, 1, true>(acc, &dh[0], fr, wp, fvoff, StoreSide<1, WT>{&dh[0], dhb, voff});
#pragma unroll
    for (int kt = 1; kt < HTM; ++kt)
      if (kt < HT)
        mm_layer_g<WT, R, 1, true>(acc, &dh[kt], fr, wp + kt * 32 * R, fvoff, StoreSide<1, WT>{&dh[kt], dhb + kt * 1024, voff});
  } else {
    store_dh();
    if (has_rgb) {
#pragma unroll
      for (int kt = 0; kt < HTM; ++kt)
        if (kt < HT) mm_chunk<WT>(acc, dh[kt], wp + kt * 32 * R, R, WT, li, h);
    }
  }
  wp += HT * 32 * R;

  // ---- trunk, last layer first.  The ReLU mask of layer l is 16 bytes of sign bits per lane, fetched one
  // layer ahead; dY_l^T is stored while the layer's MFMAs run.
  constexpr int NW = (WT + 1) / 2;
  uint32_t mcur[4] = {0u, 0u, 0u, 0u}, mnext[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int w = 0; w < NW; ++w) mcur[w] = *mask_ptr(st.mask, D - 1, h, w, Mp, m);
  for (int l = D - 1; l >= 0; --l) {
    float* dy = tile_block(st.dy + (int64_t)l * W * Mp, tix, W);
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[t][r] = ((mcur[t >> 1] >> (16 * (t & 1) + r)) & 1u) ? acc[t][r] : 0.f;
    if (l > 0) {
#pragma unroll
      for (int t = 0; t < WT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      if constexpr (kFast) {
        const BwdSide<WT> side{{g, dy, voff}, st.mask + (int64_t)((l - 1) * 8) * Mp, mnext, Mp, mvoff};
        mm_layer_g<WT, R, WT, true>(acc, g, fr, wp, fvoff, side);
      } else {
#pragma unroll
        for (int w = 0; w < NW; ++w) mnext[w] = *mask_ptr(st.mask, l - 1, h, w, Mp, m);
        store_tiles<WT>(dy, g, WT, voff);
#pragma unroll
        for (int kt = 0; kt < WT; ++kt) mm_chunk<WT>(acc, g[kt], wp + kt * 32 * R, R, WT, li, h);
      }
      wp += WT * 32 * R;
#pragma unroll
      for (int w = 0; w < 4; ++w) mcur[w] = mnext[w];
    } else {
      store_tiles<WT>(dy, g, WT, voff);
    }
  }
}

hipError_t launch_mlp_bwd_f32(const NetArgs& net, const float* d_sigma, const float* d_craw, const float* dirfeat,
                              int64_t M, int S, int has_rgb, const ActStore& st, hipStream_t stream) {
  const dim3 grid((unsigned)((M + kTile - 1) / kTile)), block(256);
  switch (net.W / 32) {
    case 1: hipLaunchKernelGGL((mlp_bwd_f32_kernel<1>), grid, block, 0, stream, net, d_sigma, d_craw, dirfeat, M, S, has_rgb, st); break;
    case 2: hipLaunchKernelGGL((mlp_bwd_f32_kernel<2>), grid, block, 0, stream, net, d_sigma, d_craw, dirfeat, M, S, has_rgb, st); break;
    case 4: hipLaunchKernelGGL((mlp_bwd_f32_kernel<4>), grid, block, 0, stream, net, d_sigma, d_craw, dirfeat, M, S, has_rgb, st); break;
    case 8: hipLaunchKernelGGL((mlp_bwd_f32_kernel<8>), grid, block, 0, stream, net, d_sigma, d_craw, dirfeat, M, S, has_rgb, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nh
