// ray_kernels.hip -- the HBM-bound per-ray kernels of the hot path (gfx950, wave64):
//   stratified sampling      nerfBlock.getSample/givePerturb/getInput (nerfBlock.scala:47-69)
//   alpha compositing fwd    nerfBlock.getWeight + weighted sum + addBkgd (nerfBlock.scala:24-45,115)
//   alpha compositing bwd    reverse mode of the same
//   inverse-CDF resampling   nerfBlock.samplePdf (nerfBlock.scala:71-106)
//   ray generation / u8      getDataSet.getRaysNp (+ndc), runNerf.renderToImage's uint8 cast
// Layout is ray-major [ray][sample]; one WAVE owns one ray and walks its samples 64 at a time, so every
// load/store is a contiguous 256-byte (768 for rgb) segment and the cumulative sums are wave-level
// prefix scans with a carry between 64-sample chunks.
//
// Exactness notes: the CDF scan runs in fp64.  The reference's cumsum (libtorch CPU)
// accumulates serially in double and rounds each element to fp32; the terms w+1e-5 lie in [2^-17, 2) so
// every partial sum of <= 4096 of them is exactly representable in double (47 significant bits needed),
// hence ANY summation order -- including this parallel scan -- yields bit-identical fp32 CDF entries
// and therefore bit-identical bin indices.
#include "common.h"
#include "rng.hip.h"

namespace nh {

// ---------------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
// sigmoid(x) = 1 / (1 + e^-x) with the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each; the argument
// scaling x*log2(e) adds |x|*1e-7 relative to e^-x): within 1e-6 relative of the correctly rounded value, against the
// 1e-5 bar of the composited colour -- and 18 VALU instructions fewer per colour channel than expf + IEEE division,
// which is what took the compositing kernels from 4.9 to 5.6 TB/s.  Saturates correctly:
// e^-x = inf -> 0, e^-x = 0 -> 1.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __expf(-x))); }

// torch.linspace(0,1,n)[i] (fp32 step; second half evaluated with one rounding)
__device__ __forceinline__ float linspace01(int i, int n) {
  if (n == 1) return 0.f;
  const float step = __fdiv_rn(1.0f, (float)(n - 1));
  return (i < n / 2) ? __fmul_rn(step, (float)i) : fmaf(-step, (float)(n - i - 1), 1.0f);
}

// ---------------------------------------------------------------------------------------------
// stratified sampling: z[ray][i]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float strat_z(float near, float far, int i, int N, int lin_disp) {
  const float t = linspace01(i, N);
  const float omt = -(__fsub_rn(t, 1.0f));
  if (lin_disp)
    return __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, near), omt), __fmul_rn(__fdiv_rn(1.0f, far), t)));
  return __fadd_rn(__fmul_rn(near, omt), __fmul_rn(far, t));
}

__global__ void stratified_kernel(const float* __restrict__ bounds, const float* __restrict__ t_rand, int64_t n, int N,
                                  int lin_disp, int perturb, uint64_t seed, uint32_t step, float* __restrict__ z) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * N) return;
  const int64_t ray = idx / N;
  const int i = (int)(idx - ray * N);
  const float near = bounds[ray * 2], far = bounds[ray * 2 + 1];
  float zi = strat_z(near, far, i, N, lin_disp);
  if (perturb) {  // givePerturb (nerfBlock.scala:50-57)
    const float zl = i > 0 ? strat_z(near, far, i - 1, N, lin_disp) : zi;
    const float zu = i < N - 1 ? strat_z(near, far, i + 1, N, lin_disp) : zi;
    const float lower = i > 0 ? __fmul_rn(__fadd_rn(zi, zl), 0.5f) : zi;
    const float upper = i < N - 1 ? __fmul_rn(__fadd_rn(zu, zi), 0.5f) : zi;
    const float r = t_rand ? t_rand[idx] : philox_uniform(seed, kStreamPerturb, step, (uint64_t)idx);
    zi = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), r));
  }
  z[idx] = zi;
}

// ---------------------------------------------------------------------------------------------
// compositing forward: one wave per ray
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ sigma,
                                                            const float* __restrict__ noise,
                                                            const float* __restrict__ craw, const float* __restrict__ z,
                                                            int64_t n, int S, float noise_std, int white,
                                                            uint64_t seed, uint32_t step, uint32_t stream,
                                                            float* __restrict__ w, float* __restrict__ rgb,
                                                            float* __restrict__ depth, float* __restrict__ acc) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n) return;
  const float* sg = sigma + ray * S;
  const float* zr = z + ray * S;
  double carry = 0.0;     // cumSum(0) carried between chunks (double accumulator, as libtorch CPU)
  float Tcarry = 1.0f;    // T of the last sample of the previous chunk
  float sw = 0.f, sz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    float a = 0.f, zi = 0.f;
    if (i < S) zi = zr[i];
    if (i < S - 1) {
      const float dist = __fsub_rn(zr[i + 1], zi);
      float s = sg[i];
      if (noise_std > 0.f) {  // addNoise (nerfBlock.scala:21)
        const float nz = noise ? noise[ray * S + i] : philox_normal(seed, stream, step, (uint64_t)(ray * S + i));
        s = __fadd_rn(s, __fmul_rn(nz, noise_std));
      }
      a = __fmul_rn(-fmaxf(s, 0.f), dist);
    }
    const double run = carry + wave_incl_scan((double)a, lane);
    const float T = expf((float)run);
    const float alpha = -(__fsub_rn(expf(a), 1.0f));
    float Tprev = __shfl_up(T, 1);
    if (lane == 0) Tprev = Tcarry;
    float wi = 0.f;
    if (i < S) wi = (i == 0) ? alpha : ((i < S - 1) ? __fmul_rn(Tprev, alpha) : Tprev);
    carry = __shfl(run, 63);
    Tcarry = __shfl(T, 63);
    if (i < S) {
      if (w) w[ray * S + i] = wi;
      sw += wi;
      sz = __fadd_rn(sz, __fmul_rn(wi, zi));
      if (craw) {
        const float* c = craw + (ray * S + i) * 3;
        c0 = __fadd_rn(c0, __fmul_rn(wi, sigmoidf_(c[0])));
        c1 = __fadd_rn(c1, __fmul_rn(wi, sigmoidf_(c[1])));
        c2 = __fadd_rn(c2, __fmul_rn(wi, sigmoidf_(c[2])));
      }
    }
  }
  sw = wave_sum(sw); sz = wave_sum(sz);
  c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
  if (lane == 0) {
    if (rgb && craw) {
      const float bk = white ? -(__fsub_rn(sw, 1.0f)) : 0.f;  // addBkgd (nerfBlock.scala:24-29)
      rgb[ray * 3 + 0] = c0 + bk;
      rgb[ray * 3 + 1] = c1 + bk;
      rgb[ray * 3 + 2] = c2 + bk;
    }
    if (depth) depth[ray] = sz;
    if (acc) acc[ray] = sw;
  }
}

// ---------------------------------------------------------------------------------------------
// compositing backward: one wave per ray; forward scan (T) then reverse scan (suffix of dT*T).
// dynamic LDS: 4 waves x 3 x Spad floats  (T, e=exp(a), dw)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ sigma,
                                                            const float* __restrict__ noise,
                                                            const float* __restrict__ craw, const float* __restrict__ z,
                                                            const float* __restrict__ d_rgb, int64_t n, int S, int Spad,
                                                            float noise_std, int white, uint64_t seed, uint32_t step,
                                                            uint32_t stream, float* __restrict__ d_sigma,
                                                            float* __restrict__ d_craw) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // wave index through readfirstlane: row pointers and the wave's LDS slice in SGPRs (see sample_pdf_kernel)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t ray_raw = (int64_t)blockIdx.x * 4 + wv;
  const bool live = ray_raw < n;
  const int64_t ray = live ? ray_raw : n - 1;
  float* Ts = smem + (size_t)wv * 3 * Spad;
  float* es = Ts + Spad;
  float* dws = es + Spad;
  const float* sg = sigma + ray * S;
  const float* zr = z + ray * S;
  const float g0 = d_rgb[ray * 3], g1 = d_rgb[ray * 3 + 1], g2 = d_rgb[ray * 3 + 2];
  const float gsum = g0 + g1 + g2;
  // pass 1 (forward): w, T, e, dw, d_craw
  double carry = 0.0;
  float Tcarry = 1.0f;
  for (int base = 0; base < S; base += 64) {
    const int i = base + lane;
    float a = 0.f, zi = 0.f;
    if (i < S) zi = zr[i];
    if (i < S - 1) {
      const float dist = __fsub_rn(zr[i + 1], zi);
      float s = sg[i];
      if (noise_std > 0.f) {
        const float nz = noise ? noise[ray * S + i] : philox_normal(seed, stream, step, (uint64_t)(ray * S + i));
        s = __fadd_rn(s, __fmul_rn(nz, noise_std));
      }
      a = __fmul_rn(-fmaxf(s, 0.f), dist);
    }
    const double run = carry + wave_incl_scan((double)a, lane);
    const float T = expf((float)run);
    const float e = expf(a);
    const float alpha = -(__fsub_rn(e, 1.0f));
    float Tprev = __shfl_up(T, 1);
    if (lane == 0) Tprev = Tcarry;
    carry = __shfl(run, 63);
    Tcarry = __shfl(T, 63);
    if (i < S) {
      const float wi = (i == 0) ? alpha : ((i < S - 1) ? __fmul_rn(Tprev, alpha) : Tprev);
      const float* c = craw + (ray * S + i) * 3;
      const float s0 = sigmoidf_(c[0]), s1 = sigmoidf_(c[1]), s2 = sigmoidf_(c[2]);
      if (live) {
        float* dc = d_craw + (ray * S + i) * 3;
        dc[0] = wi * g0 * s0 * (1.0f - s0);
        dc[1] = wi * g1 * s1 * (1.0f - s1);
        dc[2] = wi * g2 * s2 * (1.0f - s2);
      }
      const float g = g0 * s0 + g1 * s1 + g2 * s2;
      Ts[i] = T;
      es[i] = e;
      dws[i] = white ? g - gsum : g;
    }
  }
  __syncthreads();
  // pass 2 (reverse): da_j = sum_{i>=j} dT_i T_i - dalpha_j e_j ;  d sigma_j = (s_j > 0) ? -dist_j da_j : 0
  double rcarry = 0.0;
  const int nchunks = (S + 63) / 64;
  for (int cidx = nchunks - 1; cidx >= 0; --cidx) {
    const int j = cidx * 64 + (63 - lane);  // lane 0 takes the LAST sample of the chunk: forward scan == suffix sum
    double term = 0.0;
    if (j < S - 1) {
      const float dT = (j == S - 2) ? dws[S - 1] : dws[j + 1] * (1.0f - es[j + 1]);
      term = (double)dT * (double)Ts[j];
    }
    const double suf = rcarry + wave_incl_scan(term, lane);
    rcarry = __shfl(suf, 63);
    if (j < S - 1) {
      const float dalpha = dws[j] * (j == 0 ? 1.0f : Ts[j - 1]);
      const float da = (float)suf + dalpha * (-es[j]);
      const float dist = __fsub_rn(zr[j + 1], zr[j]);
      float s = sg[j];
      if (noise_std > 0.f) {
        const float nz = noise ? noise[ray * S + j] : philox_normal(seed, stream, step, (uint64_t)(ray * S + j));
        s = __fadd_rn(s, __fmul_rn(nz, noise_std));
      }
      if (live) d_sigma[ray * S + j] = s > 0.f ? (-dist) * da : 0.f;
    } else if (j == S - 1 && live) {
      d_sigma[ray * S + j] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Compositing, register form (S <= 64*K, K <= 5: every shape the reference runs -- 64, 64+64, 64+128, 64+256).
// One wave per ray; lane l owns the K CONSECUTIVE samples K*l .. K*l+K-1, so
//   * every load/store is one wide access per lane (K = 3: global_load_dwordx3; craw 36 B per lane) and the wave
//     touches each 128-B line once;
//   * the cumulative sum is K serial fp64 adds per lane plus ONE wave scan per ray, done with DPP moves
//     (row_shr 1/2/4/8 inside the 16-lane rows, row_bcast 15/31 across them): 18 VALU, no trip through the LDS
//     crossbar (the generic kernels below spend most of their time parked on ds_bpermute latencies:
//     profiles/r2_hbm_before_pmc_sq.csv, SQ_WAIT_ANY 61-66 % of wave cycles);
//   * T of the sample before a lane's first one is exp(float(exclusive prefix)): no shuffle of T;
//   * the backward pass needs no LDS at all: T, exp(a) and dw of a lane's samples stay in registers, the suffix sum
//     is a serial reverse sum per lane plus one scan over mirrored lanes.
// fp64 partial sums differ from a serial double accumulation by ~1e-16 relative (summation order), far below the
// fp32 rounding of every element.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {   // lanes that receive nothing read +0.0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ double wave_incl_scan_dpp(double v) {
  v += dpp_d<0x111, 0xf>(v);   // row_shr:1
  v += dpp_d<0x112, 0xf>(v);   // row_shr:2
  v += dpp_d<0x114, 0xf>(v);   // row_shr:4
  v += dpp_d<0x118, 0xf>(v);   // row_shr:8   -> inclusive scan inside each row of 16
  v += dpp_d<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_d<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ double wave_shr1(double v) { return dpp_d<0x138, 0xf>(v); }   // lane l <- lane l-1, lane 0 <- 0
__device__ __forceinline__ float wave_total_dpp(float v) {   // the sum of all lanes, valid in lane 63
  v += dpp_f<0x111, 0xf>(v); v += dpp_f<0x112, 0xf>(v); v += dpp_f<0x114, 0xf>(v); v += dpp_f<0x118, 0xf>(v);
  v += dpp_f<0x142, 0xa>(v); v += dpp_f<0x143, 0xc>(v);
  return v;
}
template <int K> struct __attribute__((packed, aligned(4))) PkF { float v[K]; };
// K consecutive floats of a ray row starting at sample i0 (zero past `limit`): one wide load when all are in range
template <int K>
__device__ __forceinline__ void load_run(const float* __restrict__ row, int i0, int limit, float (&out)[K]) {
  if (i0 + K <= limit) {
    const PkF<K> p = *reinterpret_cast<const PkF<K>*>(row + i0);
#pragma unroll
    for (int j = 0; j < K; ++j) out[j] = p.v[j];
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) out[j] = (i0 + j < limit) ? row[i0 + j] : 0.f;
  }
}
template <int K>
__device__ __forceinline__ void store_run(float* __restrict__ row, int i0, int limit, const float (&v)[K]) {
  if (i0 + K <= limit) {
    PkF<K> p;
#pragma unroll
    for (int j = 0; j < K; ++j) p.v[j] = v[j];
    *reinterpret_cast<PkF<K>*>(row + i0) = p;
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (i0 + j < limit) row[i0 + j] = v[j];
  }
}

// a_i = -relu(sigma_i + noise) * (z_{i+1} - z_i) of a lane's K samples (0 for i >= S-1), the effective density,
// the lane-local inclusive sums in fp64 and the exclusive prefix over the lanes before this one
template <int K>
struct LaneScan {
  float z[K + 1], seff[K], a[K];
  double cum[K], P;
};
template <int K>
__device__ __forceinline__ void lane_scan(LaneScan<K>& L, const float* __restrict__ sg, const float* __restrict__ zr,
                                          const float* __restrict__ noise_row, int i0, int S, float noise_std,
                                          uint64_t seed, uint32_t stream, uint32_t step, int64_t flat0) {
  load_run<K + 1>(zr, i0, S, L.z);
  load_run<K>(sg, i0, S - 1, L.seff);
  if (noise_std > 0.f) {  // addNoise (nerfBlock.scala:21)
    float nz[K];
    if (noise_row) load_run<K>(noise_row, i0, S - 1, nz);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (!noise_row) nz[j] = (i0 + j < S - 1) ? philox_normal(seed, stream, step, (uint64_t)(flat0 + j)) : 0.f;
      L.seff[j] = __fadd_rn(L.seff[j], __fmul_rn(nz[j], noise_std));
    }
  }
  double c = 0.0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    L.a[j] = (i0 + j < S - 1) ? __fmul_rn(-fmaxf(L.seff[j], 0.f), __fsub_rn(L.z[j + 1], L.z[j])) : 0.f;
    c += (double)L.a[j];
    L.cum[j] = c;
  }
  L.P = wave_shr1(wave_incl_scan_dpp(c));
}

template <int K>
__global__ __launch_bounds__(256) void composite_fwd_k(const float* __restrict__ sigma, const float* __restrict__ noise,
                                                       const float* __restrict__ craw, const float* __restrict__ z,
                                                       int64_t n, int S, float noise_std, int white, uint64_t seed,
                                                       uint32_t step, uint32_t stream, float* __restrict__ w,
                                                       float* __restrict__ rgb, float* __restrict__ depth,
                                                       float* __restrict__ acc) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n) return;   // wave-uniform
  const int i0 = lane * K;
  LaneScan<K> L;
  lane_scan<K>(L, sigma + ray * S, z + ray * S, noise ? noise + ray * S : nullptr, i0, S, noise_std, seed, stream, step,
               ray * S + i0);
  float cr[3 * K];
  if (craw) load_run<3 * K>(craw + ray * S * 3, 3 * i0, 3 * S, cr);
  float Tprev = expf((float)L.P);
  float wv[K], sw = 0.f, sz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int i = i0 + j;
    const float T = expf((float)(L.P + L.cum[j]));
    const float alpha = -(__fsub_rn(expf(L.a[j]), 1.0f));
    const float wi = (i == 0) ? alpha : ((i < S - 1) ? __fmul_rn(Tprev, alpha) : Tprev);
    Tprev = T;
    wv[j] = wi;
    if (i < S) {
      sw += wi;
      sz = __fadd_rn(sz, __fmul_rn(wi, L.z[j]));
      if (craw) {
        c0 = __fadd_rn(c0, __fmul_rn(wi, sigmoidf_(cr[3 * j + 0])));
        c1 = __fadd_rn(c1, __fmul_rn(wi, sigmoidf_(cr[3 * j + 1])));
        c2 = __fadd_rn(c2, __fmul_rn(wi, sigmoidf_(cr[3 * j + 2])));
      }
    }
  }
  if (w) store_run<K>(w + ray * S, i0, S, wv);
  sw = wave_total_dpp(sw); sz = wave_total_dpp(sz);
  if (craw) { c0 = wave_total_dpp(c0); c1 = wave_total_dpp(c1); c2 = wave_total_dpp(c2); }
  if (lane == 63) {
    if (rgb && craw) {
      const float bk = white ? -(__fsub_rn(sw, 1.0f)) : 0.f;  // addBkgd (nerfBlock.scala:24-29)
      rgb[ray * 3 + 0] = c0 + bk;
      rgb[ray * 3 + 1] = c1 + bk;
      rgb[ray * 3 + 2] = c2 + bk;
    }
    if (depth) depth[ray] = sz;
    if (acc) acc[ray] = sw;
  }
}

template <int K>
__global__ __launch_bounds__(256) void composite_bwd_k(const float* __restrict__ sigma, const float* __restrict__ noise,
                                                       const float* __restrict__ craw, const float* __restrict__ z,
                                                       const float* __restrict__ d_rgb, int64_t n, int S,
                                                       float noise_std, int white, uint64_t seed, uint32_t step,
                                                       uint32_t stream, float* __restrict__ d_sigma,
                                                       float* __restrict__ d_craw) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n) return;   // wave-uniform
  const int i0 = lane * K;
  const float g0 = d_rgb[ray * 3], g1 = d_rgb[ray * 3 + 1], g2 = d_rgb[ray * 3 + 2];
  const float gsum = g0 + g1 + g2;
  LaneScan<K> L;
  lane_scan<K>(L, sigma + ray * S, z + ray * S, noise ? noise + ray * S : nullptr, i0, S, noise_std, seed, stream, step,
               ray * S + i0);
  float cr[3 * K];
  load_run<3 * K>(craw + ray * S * 3, 3 * i0, 3 * S, cr);
  // forward: T, e = exp(a), dw = d(loss)/d(w) of the lane's samples; d_craw goes out
  const float Tprev0 = expf((float)L.P);
  float T[K], e[K], dw[K], dc[3 * K];
  float Tprev = Tprev0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int i = i0 + j;
    T[j] = expf((float)(L.P + L.cum[j]));
    e[j] = expf(L.a[j]);
    const float alpha = -(__fsub_rn(e[j], 1.0f));
    const float wi = (i == 0) ? alpha : ((i < S - 1) ? __fmul_rn(Tprev, alpha) : Tprev);
    Tprev = T[j];
    const float s0 = sigmoidf_(cr[3 * j]), s1 = sigmoidf_(cr[3 * j + 1]), s2 = sigmoidf_(cr[3 * j + 2]);
    dc[3 * j + 0] = wi * g0 * s0 * (1.0f - s0);
    dc[3 * j + 1] = wi * g1 * s1 * (1.0f - s1);
    dc[3 * j + 2] = wi * g2 * s2 * (1.0f - s2);
    const float g = g0 * s0 + g1 * s1 + g2 * s2;
    dw[j] = (i < S) ? (white ? g - gsum : g) : 0.f;
  }
  store_run<3 * K>(d_craw + ray * S * 3, 3 * i0, 3 * S, dc);
  // reverse: da_j = sum_{i>=j} dT_i T_i - dalpha_j e_j ;  d sigma_j = (s_j > 0) ? -dist_j da_j : 0
  const float dw_next = __shfl_down(dw[0], 1), e_next = __shfl_down(e[0], 1);   // first sample of the next lane
  double rcum[K], rc = 0.0;
#pragma unroll
  for (int j = K - 1; j >= 0; --j) {
    const int i = i0 + j;
    const float dwn = (j < K - 1) ? dw[j + 1 < K ? j + 1 : j] : dw_next;
    const float en = (j < K - 1) ? e[j + 1 < K ? j + 1 : j] : e_next;
    if (i < S - 1) {
      const float dT = (i == S - 2) ? dwn : dwn * (1.0f - en);
      rc += (double)dT * (double)T[j];
    }
    rcum[j] = rc;
  }
  // exclusive suffix over the lanes after this one: mirror the lanes, prefix-scan, mirror back
  const double mir = __shfl(rc, 63 - lane);
  const double Q = __shfl(wave_shr1(wave_incl_scan_dpp(mir)), 63 - lane);
  float ds[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const int i = i0 + j;
    const float Tm1 = (j == 0) ? Tprev0 : T[j > 0 ? j - 1 : 0];
    const float dalpha = dw[j] * (i == 0 ? 1.0f : Tm1);
    const float da = (float)(Q + rcum[j]) + dalpha * (-e[j]);
    const float dist = __fsub_rn(L.z[j + 1], L.z[j]);
    ds[j] = (i < S - 1 && L.seff[j] > 0.f) ? (-dist) * da : 0.f;
  }
  store_run<K>(d_sigma + ray * S, i0, S, ds);
}

// ---------------------------------------------------------------------------------------------
// hierarchical inverse-CDF resampling: one wave per ray, CDF and bins staged in LDS
// ---------------------------------------------------------------------------------------------
// every wave sorts its own ray's buffer: LDS operations of one wave execute in order, so a wave-level fence is all
// the synchronisation a compare-exchange stage needs (no s_barrier: waves may take different paths below)
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void bitonic_stage(float* a, int P, int lane, int k, int j) {
  for (int i = lane; i < P; i += 64) {
    const int ixj = i ^ j;
    if (ixj > i) {
      const float x = a[i], y = a[ixj];
      const bool up = (i & k) == 0;
      if ((x > y) == up) { a[i] = y; a[ixj] = x; }
    }
  }
  wave_sync_lds();
}
__device__ __forceinline__ void bitonic_sort_lds(float* a, int P, int lane) {
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) bitonic_stage(a, P, lane, k, j);
}
__device__ __forceinline__ double wave_last(double v) {   // lane 63's value, through SGPRs
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// LDS per wave: cdf[Npad] | bins[Npad] | zl[Npad] | sb[P] | ub[PU] | mb[Mpad]
__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ w, const float* __restrict__ z,
                                                         const float* __restrict__ u_rand, int64_t n, int N, int NI,
                                                         int Npad, int P, int PU, int Mpad, int perturb, uint64_t seed,
                                                         uint32_t step, float* __restrict__ cdf_out,
                                                         int32_t* __restrict__ ind_out, float* __restrict__ zs_out,
                                                         float* __restrict__ z_fine) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the wave index is wave-uniform, but only readfirstlane tells the compiler: with it the ray's row pointers and the
  // wave's LDS slice live in SGPRs and the per-lane address arithmetic is 32-bit (the kernel is VALU-issue bound, and
  // 64-bit per-lane address products were a good part of its instructions)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t ray_raw = (int64_t)blockIdx.x * 4 + wv;
  const bool live = ray_raw < n;
  const int64_t ray = live ? ray_raw : n - 1;
  // every wave works in its own slice of LDS: all synchronisation below is wave-level
  float* cdf = smem + (size_t)wv * (3 * Npad + P + PU + Mpad);
  float* bins = cdf + Npad;
  float* zl = bins + Npad;
  float* sb = zl + Npad;
  float* ub = sb + P;
  float* mb = ub + PU;
  const float* wr = w + ray * N;
  const float* zr = z + ray * N;
  const int nb = N - 1;
  // weightCum = cumsum(weight[1:-1] + 1e-5)  (nerfBlock.scala:81): fp64 scan (DPP), each element rounded to fp32
  double carry = 0.0;
  bool ordered = true;
  for (int base = 0; base < N; base += 64) {
    const int k = base + lane;
    const double t = (k < N - 2) ? (double)__fadd_rn(wr[k + 1], 1e-5f) : 0.0;
    const double run = carry + wave_incl_scan_dpp(t);
    carry = wave_last(run);
    if (k < N - 2) cdf[k + 1] = (float)run;
    if (k < N) {
      const float z0 = zr[k];
      zl[k] = z0;
      if (k < nb) {
        const float z1 = zr[k + 1];
        bins[k] = __fmul_rn(__fadd_rn(z1, z0), 0.5f);
        ordered = ordered && (z0 <= z1);
      }
    }
  }
  if (perturb) {  // randomUniform(...).sort(0)  (nerfBlock.scala:71)
    for (int m = lane; m < PU; m += 64)
      ub[m] = m < NI ? (u_rand ? u_rand[ray * NI + m] : philox_uniform(seed, kStreamPdf, step, (uint64_t)(ray * NI + m)))
                     : __builtin_inff();
  }
  wave_sync_lds();
  const float sum = cdf[N - 2];
  wave_sync_lds();
  for (int k = lane; k < nb; k += 64) cdf[k] = (k == 0) ? 0.f : __fdiv_rn(cdf[k], sum);  // cdf = [0, wc/sum]
  wave_sync_lds();
  if (perturb) bitonic_sort_lds(ub, PU, lane);
  if (cdf_out && live)
    for (int k = lane; k < nb; k += 64) cdf_out[ray * nb + k] = cdf[k];
  // The merged row sort(concat(samples, z)) (nerfBlock.scala:101) is built by RANK instead of by sorting: a sample s
  // lands at  m + #{z_i <= s},  a coarse depth z_i at  i + #{s_m < z_i}  -- the positions of a merge of two ascending
  // rows, i.e. the same multiset in the same (sorted) order as the full sort.  #{z_i <= s} is known to within one
  // step from the bin the sample fell in (bins are the midpoints of z); #{s_m < z_i} is one binary search over the
  // samples.  Rows that do not ascend (caller-supplied z through the debug entry point) take the full sort below.
  float prev_hi = -__builtin_inff();
  for (int base = 0; base < NI; base += 64) {
    const int m = base + lane;
    float s = __builtin_inff();
    if (m < NI) {
      const float u = perturb ? ub[m] : linspace01(m, NI);
      // id = #{cdf <= u}: rank by (stable) double argsort minus m (nerfBlock.scala:89-90).  (A fixed-step, branch-free
      // counting search was measured in round 3: 14 % slower than this lo/hi loop.)
      int lo = 0, hi = nb;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
      }
      const int id = lo;
      const int below = max(id - 1, 0);
      const int above = min(id, nb - 1);
      const float cb = cdf[below], bb = bins[below];
      float den = __fsub_rn(cdf[above], cb);
      if (den < 1e-5f) den = 1.0f;
      const float t = __fdiv_rn(__fsub_rn(u, cb), den);
      s = __fadd_rn(bb, __fmul_rn(t, __fsub_rn(bins[above], bb)));
      sb[m] = s;
      if (live) {
        if (ind_out) ind_out[ray * NI + m] = id;
        if (zs_out) zs_out[ray * NI + m] = s;
      }
      int g = min(below + 2, N);   // #{z_i <= s}: z_{below+1} is the only depth inside [bins[below], bins[above]]
      while (g < N && zl[g] <= s) ++g;
      while (g > 0 && zl[g - 1] > s) --g;
      mb[m + g] = s;
    }
    // samples ascend? (inside this chunk of 64 and against the chunk before)
    float left = __shfl_up(s, 1);
    if (lane == 0) left = prev_hi;
    ordered = ordered && (m >= NI || left <= s);
    prev_hi = __shfl(s, 63);
  }
  wave_sync_lds();
  for (int i = lane; i < N; i += 64) {
    const float zi = zl[i];
    int lo = 0, hi = NI;   // #{s_m < z_i}
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sb[mid] < zi) lo = mid + 1; else hi = mid;
    }
    mb[i + lo] = zi;
  }
  ordered = __all(ordered);
  wave_sync_lds();
  const float* src = mb;
  if (!ordered) {
    for (int i = lane; i < P - NI; i += 64) sb[NI + i] = (i < N) ? zl[i] : __builtin_inff();
    wave_sync_lds();
    bitonic_sort_lds(sb, P, lane);
    src = sb;
  }
  if (live)
    for (int i = lane; i < N + NI; i += 64) z_fine[ray * (N + NI) + i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// ray generation (getDataSet.scala:12-21,105-112; llff.scala:96-132) and uint8 conversion
// ---------------------------------------------------------------------------------------------
struct Pose { float m[12]; };

// rows [row0, row0 + nrows) of the H x Wd image; output index p = (j - row0) * Wd + i (a rank's row block of a sharded frame
// starts at the head of its own buffers)
__global__ void gen_rays_kernel(int H, int Wd, float focal, Pose c2w, float near, float far, int use_ndc, int row0, int nrows,
                                float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ bounds) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (int64_t)nrows * Wd) return;
  const int jl = (int)(p / Wd), i = (int)(p - (int64_t)jl * Wd);
  const int j = row0 + jl;
  const float halfW = (float)(Wd * .5), halfH = (float)(H * .5);
  const float dir0 = __fdiv_rn(__fsub_rn((float)i, halfW), focal);
  const float dir1 = -__fdiv_rn(__fsub_rn((float)j, halfH), focal);
  const float dir2 = -1.0f;
  float d[3], o[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    d[k] = __fadd_rn(__fadd_rn(__fmul_rn(dir0, c2w.m[k * 4 + 0]), __fmul_rn(dir1, c2w.m[k * 4 + 1])),
                     __fmul_rn(dir2, c2w.m[k * 4 + 2]));
    o[k] = c2w.m[k * 4 + 3];
  }
  if (use_ndc) {
    const float nr = 1.0f;
    const float t = __fdiv_rn(__fadd_rn(o[2], nr), d[2]);
    const float o0 = __fsub_rn(o[0], __fmul_rn(t, d[0])), o1 = __fsub_rn(o[1], __fmul_rn(t, d[1])),
                o2 = __fsub_rn(o[2], __fmul_rn(t, d[2]));
    const float sx = __fdiv_rn(-1.0f, __fdiv_rn((float)Wd, __fmul_rn(2.0f, focal)));
    const float sy = __fdiv_rn(-1.0f, __fdiv_rn((float)H, __fmul_rn(2.0f, focal)));
    const float n0 = __fmul_rn(__fdiv_rn(o0, o2), sx), n1 = __fmul_rn(__fdiv_rn(o1, o2), sy);
    const float n2 = __fadd_rn(__fdiv_rn(2.0f * nr, o2), 1.0f);
    const float e0 = __fmul_rn(__fsub_rn(__fdiv_rn(d[0], d[2]), __fdiv_rn(o0, o2)), sx);
    const float e1 = __fmul_rn(__fsub_rn(__fdiv_rn(d[1], d[2]), __fdiv_rn(o1, o2)), sy);
    const float e2 = __fdiv_rn(-2.0f * nr, o2);
    o[0] = n0; o[1] = n1; o[2] = n2;
    d[0] = e0; d[1] = e1; d[2] = e2;
  }
  const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rays_o[p * 3 + k] = o[k];
    rays_d[p * 3 + k] = __fdiv_rn(d[k], nrm);
  }
  bounds[p * 2] = use_ndc ? 0.f : __fmul_rn(nrm, near);
  bounds[p * 2 + 1] = use_ndc ? nrm : __fmul_rn(nrm, far);
}

// the library's counter-based random streams, written out (parity tests: the implicit-random branches of the kernels
// above must equal the explicit-tensor branches fed with these values)
__global__ void random_kernel(uint64_t seed, uint32_t stream, uint32_t step, int64_t n, int normal, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = normal ? philox_normal(seed, stream, step, (uint64_t)i) : philox_uniform(seed, stream, step, (uint64_t)i);
}

__global__ void to_u8_kernel(const float* __restrict__ rgb, int64_t n, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = __fmul_rn(rgb[i], 255.0f);
  v = fminf(fmaxf(v, 0.f), 255.f);
  out[i] = (uint8_t)v;  // truncation, as NDArray.toType(UINT8)
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// dynamic LDS of the two per-ray kernels that stage a ray per wave (4 waves per workgroup); nerfhip_create rejects
// configurations that do not fit 64 KiB
size_t composite_bwd_lds_bytes(int S) { return (size_t)4 * 3 * ((S + 3) & ~3) * sizeof(float); }
size_t sample_pdf_lds_bytes(int N, int NI, int perturb) {
  return (size_t)4 * (3 * ((N + 3) & ~3) + next_pow2(N + NI) + (perturb ? next_pow2(NI) : 0) + ((N + NI + 3) & ~3)) * sizeof(float);
}

hipError_t launch_stratified(const float* bounds, const float* t_rand, int64_t n, int N, int lin_disp, int perturb,
                             uint64_t seed, uint32_t step, float* z, hipStream_t s) {
  const int64_t total = n * N;
  hipLaunchKernelGGL(stratified_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, bounds, t_rand, n, N,
                     lin_disp, perturb, seed, step, z);
  return hipGetLastError();
}

hipError_t launch_composite_fwd(const float* sigma, const float* noise, const float* craw, const float* z, int64_t n,
                                int S, float noise_std, int white, uint64_t seed, uint32_t step, uint32_t stream,
                                float* w, float* rgb, float* depth, float* acc, hipStream_t s) {
  const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define NH_FWD_K(KK) hipLaunchKernelGGL(composite_fwd_k<KK>, grid, block, 0, s, sigma, noise, craw, z, n, S, noise_std, white, \
                                        seed, step, stream, w, rgb, depth, acc)
  switch ((S + 63) / 64) {   // samples per lane of the register form; longer rays take the generic chunked kernel
    case 1: NH_FWD_K(1); break;
    case 2: NH_FWD_K(2); break;
    case 3: NH_FWD_K(3); break;
    case 4: NH_FWD_K(4); break;
    case 5: NH_FWD_K(5); break;
    default:
      hipLaunchKernelGGL(composite_fwd_kernel, grid, block, 0, s, sigma, noise, craw, z, n, S, noise_std, white, seed, step,
                         stream, w, rgb, depth, acc);
  }
#undef NH_FWD_K
  return hipGetLastError();
}

hipError_t launch_composite_bwd(const float* sigma, const float* noise, const float* craw, const float* z,
                                const float* d_rgb, int64_t n, int S, float noise_std, int white, uint64_t seed,
                                uint32_t step, uint32_t stream, float* d_sigma, float* d_craw, hipStream_t s) {
  const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define NH_BWD_K(KK) hipLaunchKernelGGL(composite_bwd_k<KK>, grid, block, 0, s, sigma, noise, craw, z, d_rgb, n, S, noise_std, \
                                        white, seed, step, stream, d_sigma, d_craw)
  switch ((S + 63) / 64) {
    case 1: NH_BWD_K(1); return hipGetLastError();
    case 2: NH_BWD_K(2); return hipGetLastError();
    case 3: NH_BWD_K(3); return hipGetLastError();
    case 4: NH_BWD_K(4); return hipGetLastError();
    case 5: NH_BWD_K(5); return hipGetLastError();
    default: break;
  }
#undef NH_BWD_K
  const int Spad = (S + 3) & ~3;
  const size_t lds = composite_bwd_lds_bytes(S);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), lds, s, sigma, noise, craw, z,
                     d_rgb, n, S, Spad, noise_std, white, seed, step, stream, d_sigma, d_craw);
  return hipGetLastError();
}

hipError_t launch_sample_pdf(const float* w, const float* z, const float* u_rand, int64_t n, int N, int NI, int perturb,
                             uint64_t seed, uint32_t step, float* cdf, int32_t* ind, float* zs, float* z_fine,
                             hipStream_t s) {
  const int Npad = (N + 3) & ~3;
  const int P = next_pow2(N + NI);
  const int PU = perturb ? next_pow2(NI) : 0;
  const size_t lds = sample_pdf_lds_bytes(N, NI, perturb);
  if (lds > 64 * 1024 || N < 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), lds, s, w, z, u_rand, n, N, NI, Npad,
                     P, PU, (N + NI + 3) & ~3, perturb, seed, step, cdf, ind, zs, z_fine);
  return hipGetLastError();
}

hipError_t launch_gen_rays(int H, int Wd, float focal, const float* c2w, float near, float far, int use_ndc, int row0,
                           int nrows, float* rays_o, float* rays_d, float* bounds, hipStream_t s) {
  if (row0 < 0 || nrows < 0 || row0 + nrows > H) return hipErrorInvalidValue;
  if (nrows == 0) return hipSuccess;
  Pose p;
  for (int i = 0; i < 12; ++i) p.m[i] = c2w[i];
  const int64_t total = (int64_t)nrows * Wd;
  hipLaunchKernelGGL(gen_rays_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, H, Wd, focal, p, near,
                     far, use_ndc, row0, nrows, rays_o, rays_d, bounds);
  return hipGetLastError();
}

hipError_t launch_random(uint64_t seed, uint32_t stream, uint32_t step, int64_t n, int normal, float* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(random_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seed, stream, step, n, normal, out);
  return hipGetLastError();
}

hipError_t launch_to_u8(const float* rgb, int64_t n, uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(to_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rgb, n, out);
  return hipGetLastError();
}

}  // namespace nh
