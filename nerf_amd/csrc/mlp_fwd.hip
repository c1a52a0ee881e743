// mlp_fwd.hip -- fused positional-encoding + density/colour MLP forward on fp32 MFMA (gfx950).
//
// Computes nerfCoreBlock.forwardInternal (reference nerf/nerfCoreBlock.scala:107-112) for a flat list of
// samples m = ray*S + i: position o + d*z (nerfBlock.scala:65), positionCode (:127-136) generated in
// registers as the B operand of layer 0 (and of the layer after a skip), D x (Linear+ReLU) with the
// skip concat (:19-32), density head (:34-44), colour head view-dir / SH / plain (:50-85).
// No activation ever leaves the CU in inference; the training variant also stores the post-ReLU
// activations feature-major for the backward pass.
//
// Work decomposition: one workgroup = 4 waves = 128 samples; one wave = 32 samples x all features;
// 1 wave per SIMD with the 512-register file (accumulators 128 + previous layer 128 + encoding 32).
// Weights: every wave reads its 32x32 fragments straight from the packed stream in L2 through a ring of
// prefetched register sets that runs continuously across layers (mlp_common.hip.h, "direct-from-L2"); the
// training variant interleaves the activation stores with those loads.
#include "mlp_common.hip.h"

namespace nh {

template <int WT, bool TRAIN>
__global__ __launch_bounds__(256, 1) void mlp_fwd_f32_kernel(NetArgs net, const float* __restrict__ rays_o,
                                                             const float* __restrict__ rays_d,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ dirfeat, int64_t M, int S,
                                                             float* __restrict__ sigma, float* __restrict__ craw,
                                                             int want_rgb, ActStore st) {
  constexpr int R = 32 * WT;
  constexpr int HTM = WT >= 2 ? WT / 2 : 1;
  constexpr bool kFast = WT >= 4;   // continuous fragment ring needs tile counts that are multiples of kNB
  __shared__ __attribute__((aligned(16))) float sbias[kMaxD * R];   // biases: LDS reads do not touch vmcnt

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int64_t m = (int64_t)blockIdx.x * kTile + wave * 32 + li;
  const bool valid = m < M;
  const int64_t mc = valid ? m : M - 1;
  const int64_t ray = mc / S;
  const float* vp = net.vecpack;
  const int W = net.W, D = net.D;
  const int64_t Mp = TRAIN ? st.Mpad : 0;
  const uint32_t voff = tile_voff(lane);                                        // activation-store lane offset
  const int tix = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave);   // this wave's 32-sample tile

  const float* wp = (const float*)net.wpack_fwd;   // packed weight stream, consumed front to back
  const uint32_t fvoff = frag_voff(lane);
  f32x4 fr[kNB][4];
  for (int i = tid; i < D * R; i += 256) sbias[i] = vp[i];

  // sample position (raysO.add(raysD.mul(zVals)), nerfBlock.scala:65): separate mul and add
  const float zz = z[mc];
  const float px = __fadd_rn(rays_o[ray * 3 + 0], __fmul_rn(rays_d[ray * 3 + 0], zz));
  const float py = __fadd_rn(rays_o[ray * 3 + 1], __fmul_rn(rays_d[ray * 3 + 1], zz));
  const float pz = __fadd_rn(rays_o[ray * 3 + 2], __fmul_rn(rays_d[ray * 3 + 2], zz));

  // Rows past M repeat sample M-1 (finite values); their gradients are zero in the backward pass, so the
  // stored activations of those columns never contribute to a weight gradient.
  f32x16 enc[2];
  if (trig_reduced(px, py, pz, net.E)) {   // wave-uniform: the usual case, sin/cos by the short routine (mlp_common.hip.h)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) enc[t][r] = code_value_h<true>(px, py, pz, tile_row(t, r, 0), h, net.E);
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) enc[t][r] = code_value_h<false>(px, py, pz, tile_row(t, r, 0), h, net.E);
  }
  __syncthreads();

  f32x16 act[WT], acc[WT];
  auto store_mask = [&](int slot) {
    uint32_t mw[4];
    relu_mask_words<WT>(act, mw, WT);
#pragma unroll
    for (int w = 0; w < (WT + 1) / 2; ++w) *mask_ptr(st.mask, slot, h, w, st.Mpad, m) = mw[w];
  };

  auto relu_to_act = [&]() {
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) act[t][r] = fmaxf(acc[t][r], 0.f);
  };
  // layer 0: encoding only.  The fragment ring is primed directly in front of its first part: asm loads are
  // only ever in flight inside a part (mlp_common.hip.h).
  load_rows<WT>(acc, sbias, WT, W, h);  // accumulator := bias
  if constexpr (kFast) {
    prime_frags(fr, wp, fvoff);
    if constexpr (TRAIN) mm_layer_g<WT, R, 2>(acc, enc, fr, wp, fvoff, StoreSide<2, 2 * WT>{enc, tile_block(st.enc, tix, 64), voff});
    else mm_layer_g<WT, R, 2>(acc, enc, fr, wp, fvoff, NoSide{});
  } else {
    if (TRAIN) store_tiles<2>(tile_block(st.enc, tix, 64), enc, 2, voff);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) mm_chunk<WT>(acc, enc[kt], wp + kt * 32 * R, R, WT, li, h);
  }
  wp += 2 * 32 * R;
  relu_to_act();
  for (int l = 1; l < D; ++l) {
    load_rows<WT>(acc, sbias + l * R, WT, W, h);
    float* dst = TRAIN ? tile_block(st.act + (int64_t)(l - 1) * W * Mp, tix, W) : nullptr;
    if (TRAIN) store_mask(l - 1);
    if constexpr (kFast) {
      // TRAIN: the post-ReLU activations of layer l-1 (the B operand of this layer) go out two rows per tile
      // step, between the fragment loads, so the HBM writes overlap the MFMAs of the whole layer
      if constexpr (TRAIN) mm_layer_g<WT, R, WT>(acc, act, fr, wp, fvoff, StoreSide<WT, WT * WT>{act, dst, voff});
      else mm_layer_g<WT, R, WT>(acc, act, fr, wp, fvoff, NoSide{});
    } else {
      if (TRAIN) store_tiles<WT>(dst, act, WT, voff);
#pragma unroll
      for (int kt = 0; kt < WT; ++kt) mm_chunk<WT>(acc, act[kt], wp + kt * 32 * R, R, WT, li, h);
    }
    wp += WT * 32 * R;
    if ((net.skip_mask >> (l - 1)) & 1u) {
      if constexpr (kFast) {
        mm_layer_g<WT, R, 2>(acc, enc, fr, wp, fvoff, NoSide{});
      } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) mm_chunk<WT>(acc, enc[kt], wp + kt * 32 * R, R, WT, li, h);
      }
      wp += 2 * 32 * R;
    }
    relu_to_act();
  }
  if (TRAIN) store_mask(D - 1);
  if (TRAIN && !want_rgb) store_tiles<WT>(tile_block(st.act + (int64_t)(D - 1) * W * Mp, tix, W), act, WT, voff);

  // density head (nerfCoreBlock.scala:39-40): dot over this lane's half of the features + the other half
  {
    const float* ws = vp + D * W;
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + 32 * t + 8 * q + 4 * h);
        part = fmaf(act[t][4 * q + 0], wv[0], part);
        part = fmaf(act[t][4 * q + 1], wv[1], part);
        part = fmaf(act[t][4 * q + 2], wv[2], part);
        part = fmaf(act[t][4 * q + 3], wv[3], part);
      }
    part += __shfl_xor(part, 32);
    if (valid && h == 0) sigma[m] = part + vp[(D + 5) * W];
  }
  if (!want_rgb) return;

  // colour head (nerfCoreBlock.scala:50-89)
  const int HT = net.HT;
  const int Rh = 32 * HT;
  f32x16 hacc[HTM];
  if (net.mode == 1) load_rows<HTM>(hacc, dirfeat + ray * Rh, HT, Rh, h);   // Wd.enc(dir) + bd + bf, per ray
  else load_rows<HTM>(hacc, vp + (D + 1) * W, HT, Rh, h);                   // head bias (zero padded)
  {
    float* dst = TRAIN ? tile_block(st.act + (int64_t)(D - 1) * W * Mp, tix, W) : nullptr;
    if (kFast && HT == HTM) {
      if constexpr (kFast) {
        constexpr int RH = 32 * HTM;
        const uint32_t hvoff = frag_voff(lane);
        prime_frags(fr, wp, hvoff);
        if constexpr (TRAIN) mm_layer_g<HTM, RH, WT>(hacc, act, fr, wp, hvoff, StoreSide<WT, WT * HTM>{act, dst, voff});
        else mm_layer_g<HTM, RH, WT>(hacc, act, fr, wp, hvoff, NoSide{});
      }
    } else {
      if (TRAIN) store_tiles<WT>(dst, act, WT, voff);
#pragma unroll
      for (int kt = 0; kt < WT; ++kt) mm_chunk<HTM>(hacc, act[kt], wp + kt * 32 * Rh, Rh, HT, li, h);
    }
  }

  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  if (net.mode == 1) {
    const float* wc = vp + (D + 2) * W;  // [3][Rh] zero padded
#pragma unroll
    for (int t = 0; t < HTM; ++t) {
      if (t < HT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[t][r] = fmaxf(hacc[t][r], 0.f);  // add().relu() (:67)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 32 * t + 8 * q + 4 * h;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wc + row);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(wc + Rh + row);
          const f32x4 w2 = *reinterpret_cast<const f32x4*>(wc + 2 * Rh + row);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            c0 = fmaf(hacc[t][4 * q + j], w0[j], c0);
            c1 = fmaf(hacc[t][4 * q + j], w1[j], c1);
            c2 = fmaf(hacc[t][4 * q + j], w2[j], c2);
          }
        }
      }
    }
    if (TRAIN) {
      store_tiles<HTM>(tile_block(st.hc, tix, Rh), hacc, HT, voff);
      uint32_t mw[4];
      relu_mask_words<HTM>(hacc, mw, HT);
#pragma unroll
      for (int w = 0; w < (HTM + 1) / 2; ++w) *mask_ptr(st.mask, D, h, w, st.Mpad, m) = mw[w];
    }
    if (h == 0) {
      c0 += vp[(D + 5) * W + 1];
      c1 += vp[(D + 5) * W + 2];
      c2 += vp[(D + 5) * W + 3];
    }
  } else if (net.mode == 2) {
    // reshape(...,9,3).mul(SH).sum(-2)  (:55): c[k] = sum_j q[3j+k] * Y_j(dir)
    const float* Y = dirfeat + ray * 12;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile_row(0, r, h);
      if (row < 27) {
        const int j = row / 3, k = row - 3 * j;
        const float v = hacc[0][r] * Y[j];
        c0 += k == 0 ? v : 0.f;
        c1 += k == 1 ? v : 0.f;
        c2 += k == 2 ? v : 0.f;
      }
    }
  } else {
    if (h == 0) { c0 = hacc[0][0]; c1 = hacc[0][1]; c2 = hacc[0][2]; }
  }
  c0 += __shfl_xor(c0, 32);
  c1 += __shfl_xor(c1, 32);
  c2 += __shfl_xor(c2, 32);
  if (valid && h == 0) {
    craw[m * 3 + 0] = c0;
    craw[m * 3 + 1] = c1;
    craw[m * 3 + 2] = c2;
  }
}

template <int WT>
static hipError_t launch_wt(bool train, const NetArgs& net, const float* rays_o, const float* rays_d, const float* z,
                            const float* dirfeat, int64_t M, int S, float* sigma, float* craw, int want_rgb,
                            const ActStore& st, hipStream_t stream) {
  const dim3 grid((unsigned)((M + kTile - 1) / kTile)), block(256);
  if (train)
    hipLaunchKernelGGL((mlp_fwd_f32_kernel<WT, true>), grid, block, 0, stream, net, rays_o, rays_d, z, dirfeat, M, S,
                       sigma, craw, want_rgb, st);
  else
    hipLaunchKernelGGL((mlp_fwd_f32_kernel<WT, false>), grid, block, 0, stream, net, rays_o, rays_d, z, dirfeat, M, S,
                       sigma, craw, want_rgb, st);
  return hipGetLastError();
}

hipError_t launch_mlp_fwd_f32(bool train, const NetArgs& net, const float* rays_o, const float* rays_d, const float* z,
                              const float* dirfeat, int64_t M, int S, float* sigma, float* craw, int want_rgb,
                              const ActStore& st, hipStream_t stream) {
  switch (net.W / 32) {
    case 1: return launch_wt<1>(train, net, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, st, stream);
    case 2: return launch_wt<2>(train, net, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, st, stream);
    case 4: return launch_wt<4>(train, net, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, st, stream);
    case 8: return launch_wt<8>(train, net, rays_o, rays_d, z, dirfeat, M, S, sigma, craw, want_rgb, st, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace nh
