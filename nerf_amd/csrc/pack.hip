// pack.hip -- re-lays the canonical fp32 parameters (DJL order, W[out][in]) into the streams the fused MLP
// kernels consume, the per-ray view-direction branch of the colour head, Adam and the L2 loss.
//   * weight chunks: 32 k-values x R rows, image [g = k/4][i][k%4] (see mlp_common.hip.h); the forward
//     stream holds W_l, the backward stream W_l^T (so both are the A operand of the same MFMA loop);
//   * vecpack: 16-byte aligned copies of biases / density row / colour output rows.
// Runs once per set_params and once after every optimizer step (about 2x the parameter bytes).
#include "launchers.h"
#include "rng.hip.h"

namespace nh {

__global__ void pack_chunks_kernel(const float* __restrict__ params, const ChunkDesc* __restrict__ descs,
                                   float* __restrict__ dst) {
  const ChunkDesc d = descs[blockIdx.x];
  const float* src = params + d.src_off;
  const int total = 32 * d.R;
  // tile-linear image (mlp_common.hip.h): e = (((ot*4 + q)*64 + h*32 + li)*4 + j)  <->  row i = 32 ot + li,
  // k = 8q + 4h + j: one output tile of the chunk = 4 KiB contiguous = four 1-KiB wave loads
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int j = e & 3, lane = (e >> 2) & 63, q = (e >> 8) & 3, ot = e >> 10;
    const int i = 32 * ot + (lane & 31);
    const int k = 8 * q + 4 * (lane >> 5) + j;
    float v = 0.f;
    if (d.c0 >= 0 && k < d.kvalid && i < d.ivalid)
      v = d.transposed ? src[(int64_t)(d.c0 + k) * d.ld + i] : src[(int64_t)i * d.ld + d.c0 + k];
    dst[d.dst_off + e] = v;
  }
}

// bf16 weight streams: one block (64 lanes) per fragment; lane (h, i) writes 8 bf16 = W[row0+i][col0 + F(s,h,e)]
__device__ __forceinline__ void pack_frag_bf16(const float* __restrict__ params, const FragDesc& d, __bf16* __restrict__ dst,
                                               int frag, int lane) {
  const float* src = params + (d.src_off < 0 ? 0 : d.src_off);
  __bf16* o = dst + ((int64_t)frag * 64 + lane) * 8;
  if (d.transposed == 2) {
    // A operand of v_mfma_f32_16x16x32_bf16 (mlp_fwd_bf16_t16.hip): 16 rows x the 32 features of the k-tile, lane (c, g)
    // holds row row0 + c, k-slots 8 g .. 8 g + 7 = features 16 (e >> 2) + 4 g + (e & 3)
    const int g = lane >> 4, row = d.row0 + (lane & 15);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 16 * (e >> 2) + 4 * g + (e & 3);
      float v = 0.f;
      if (d.src_off >= 0 && row < d.nrows && f < d.kvalid) v = src[(int64_t)row * d.ld + d.col0 + f];
      o[e] = (__bf16)v;
    }
    return;
  }
  const int h = lane >> 5, i = lane & 31;
  const int row = d.row0 + i;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int f = 16 * d.s + (e & 3) + 8 * (e >> 2) + 4 * h;   // bf16_common.hip.h: bf_feature
    float v = 0.f;
    if (d.src_off >= 0 && row < d.nrows && f < d.kvalid)
      v = d.transposed ? src[(int64_t)(d.col0 + f) * d.ld + row] : src[(int64_t)row * d.ld + d.col0 + f];
    o[e] = (__bf16)v;
  }
}
// split-f16 weight streams (f16x3_common.hip.h): descriptor i -> fragment 2i = hi = f16(64 w), fragment 2i+1 = lo =
// f16(64 w - hi); same lane / element order as the bf16 fragments
// range guard: an element with |64 w| >= 65504 has no f16 high half (it would become inf); its flat index + 1 goes into
// *flag (largest offender wins), nerfhip_set_params / the next synchronising call turn that into NERFHIP_E_INVALID
__global__ __launch_bounds__(64) void pack_frags_f16x3_kernel(const float* __restrict__ params,
                                                              const FragDesc* __restrict__ descs,
                                                              _Float16* __restrict__ dst, uint32_t* __restrict__ flag) {
  const FragDesc d = descs[blockIdx.x];
  const int lane = threadIdx.x, h = lane >> 5, i = lane & 31;
  const float* src = params + (d.src_off < 0 ? 0 : d.src_off);
  _Float16* oh = dst + ((int64_t)(2 * blockIdx.x) * 64 + lane) * 8;
  _Float16* ol = oh + 512;
  const bool t16 = d.transposed == 2;   // A operand order of the 16x16x32 MFMA (pack_frag_bf16; mlp_fwd_f16x3_t16.hip)
  const int row = d.row0 + (t16 ? (lane & 15) : i);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int f = t16 ? 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3) : 16 * d.s + (e & 3) + 8 * (e >> 2) + 4 * h;
    float v = 0.f;
    int64_t at = -1;
    if (d.src_off >= 0 && row < d.nrows && f < d.kvalid) {
      at = d.transposed == 1 ? (int64_t)(d.col0 + f) * d.ld + row : (int64_t)row * d.ld + d.col0 + f;
      v = src[at];
    }
    const float vs = __fmul_rn(v, 64.0f);   // exact
    if (!(fabsf(vs) < 65504.0f) && at >= 0 && flag) atomicMax(flag, (uint32_t)(d.src_off + at) + 1u);   // also catches NaN
    const _Float16 hi = (_Float16)vs;
    oh[e] = hi;
    ol[e] = (_Float16)(__fsub_rn(vs, (float)hi));
  }
}

// bf16 path vectors (fp32): [D*W] trunk biases | [(HT+1)*32] head accumulator init (mode 1: zero, the per-ray
// direction vector carries bf + bd; else bf; tile HT: b_sigma at row 0) | [32] colour bias (mode 1)
__device__ __forceinline__ void pack_bvec_range(const float* __restrict__ params, const NetLayout& L, int HT,
                                                float* __restrict__ bv, int first, int stride, float scale = 1.0f) {
  const int W = L.W, D = L.D;
  const int total = D * W + (HT + 1) * 32 + 32;
  for (int idx = first; idx < total; idx += stride) {
    float v = 0.f;
    if (idx < D * W) {
      v = params[L.b_off[idx / W] + idx % W];
    } else if (idx < D * W + HT * 32) {
      const int j = idx - D * W;
      if (L.mode != 1 && j < L.head_rows) v = params[L.bf_off + j];
    } else if (idx < D * W + (HT + 1) * 32) {
      if (idx == D * W + HT * 32) v = params[L.bs_off];
    } else {
      const int j = idx - (D * W + (HT + 1) * 32);
      if (L.mode == 1 && j < 3) v = params[L.bc_off + j];
    }
    bv[idx] = v * scale;
  }
}
__global__ void pack_bvec_kernel(const float* __restrict__ params, NetLayout L, int HT, float* __restrict__ bv, float scale) {
  pack_bvec_range(params, L, HT, bv, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, scale);
}

// Everything the bf16 kernels read of the parameters, in ONE launch (it runs after every optimizer step, and eight
// separate launches cost eight dispatch latencies of ~4.7 us for 15 us of work): blocks [0, nfrag) pack one fragment
// each of the segments laid end to end (forward / density-only / backward stream of every network), the kBvecBlocks
// blocks per network behind them write the fp32 vectors.
__global__ __launch_bounds__(64) void repack_bf16_kernel(RepackBf16Args a) {
  const int b = blockIdx.x;
  if (b < a.nfrag) {
    int sgi = 0;
#pragma unroll
    for (int k = 1; k < RepackBf16Args::kMaxSeg; ++k)
      if (k < a.nseg && b >= a.seg[k].first) sgi = k;
    const RepackBf16Args::Seg& sg = a.seg[sgi];
    pack_frag_bf16(sg.params, sg.descs[b - sg.first], (__bf16*)sg.dst, b - sg.first, threadIdx.x);
  } else {
    const int net = (b - a.nfrag) / RepackBf16Args::kBvecBlocks, blk = (b - a.nfrag) % RepackBf16Args::kBvecBlocks;
    pack_bvec_range(a.vparams[net], a.L[net], a.HT[net], a.bv[net], blk * 64 + threadIdx.x, RepackBf16Args::kBvecBlocks * 64);
  }
}

__global__ void pack_vec_kernel(const float* __restrict__ params, NetLayout L, int HT, float* __restrict__ vp) {
  const int W = L.W, D = L.D, Rh = 32 * HT;
  const int total = (D + 5) * W + 4;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (idx < D * W) {
      const int l = idx / W, j = idx - l * W;
      v = params[L.b_off[l] + j];
    } else if (idx < (D + 1) * W) {
      v = params[L.ws_off + (idx - D * W)];
    } else if (idx < (D + 2) * W) {
      const int j = idx - (D + 1) * W;
      v = j < L.head_rows ? params[L.bf_off + j] : 0.f;
    } else if (idx < (D + 5) * W) {
      const int t = idx - (D + 2) * W;
      const int k = t / Rh, j = t - k * Rh;
      v = (L.mode == 1 && k < 3 && j < L.H2) ? params[L.wc_off + (int64_t)k * L.H2 + j] : 0.f;
    } else {
      const int t = idx - (D + 5) * W;
      if (t == 0) v = params[L.bs_off];
      else v = (L.mode == 1) ? params[L.bc_off + t - 1] : 0.f;
    }
    vp[idx] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// per-ray direction features
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dir_code(const float* v, int e) {  // positionCode(dir, dirL)[e]
  if (e < 3) return v[e];
  const int t = e - 3, f = t / 6, rem = t - 6 * f;
  const int c = rem >= 3 ? rem - 3 : rem;
  const float a = v[c] * (float)(1 << f);
  return rem >= 3 ? cosf(a) : sinf(a);
}

// mode 1: dirvec[ray][Rh] = Wd . positionCode(viewdir, dirL) + bd + bf  (the Linear(W/2) on enc(dir) of
// nerfCoreBlock.scala:72-74 plus both biases of the add at :67); rows >= H2 are 0.
// Optionally writes enc(dir)^T, tile-major [Bpad/32][32][32] (common.h), for the weight gradient.
__global__ __launch_bounds__(256) void dir_branch_kernel(const float* __restrict__ params, NetLayout L, int Rh,
                                                         const float* __restrict__ viewdir, int64_t n,
                                                         float* __restrict__ dirvec, float* __restrict__ encT,
                                                         int64_t Bpad) {
  __shared__ float enc[8][32];
  const int rpb = 256 / Rh;  // rays per block (Rh in {32,64,128})
  const int64_t ray0 = (int64_t)blockIdx.x * rpb;
  const int Ed = L.Ed;
  for (int t = threadIdx.x; t < rpb * 32; t += 256) {
    const int r = t >> 5, e = t & 31;
    const int64_t ray = ray0 + r;
    float v = 0.f;
    if (ray < n && e < Ed) v = dir_code(viewdir + ray * 3, e);
    enc[r][e] = v;
    if (encT && ray < n) encT[tiled_index(e, ray, 32)] = v;
  }
  __syncthreads();
  const int r = threadIdx.x / Rh, j = threadIdx.x - r * Rh;
  const int64_t ray = ray0 + r;
  if (r >= rpb || ray >= n) return;
  float acc = 0.f;
  if (j < L.H2) {
    const float* wd = params + L.wd_off + (int64_t)j * Ed;
    acc = params[L.bd_off + j];
    for (int e = 0; e < Ed; ++e) acc = __fadd_rn(acc, __fmul_rn(enc[r][e], wd[e]));
    acc = __fadd_rn(acc, params[L.bf_off + j]);
  }
  dirvec[ray * Rh + j] = acc;
}

// mode 2: SH2(viewdir) (nerfCoreBlock.scala:138-174) -> Y[ray][12] (9 used)
__global__ void sh2_kernel(const float* __restrict__ viewdir, int64_t n, float* __restrict__ Y) {
  const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n) return;
  const float x = viewdir[ray * 3], y = viewdir[ray * 3 + 1], z = viewdir[ray * 3 + 2];
  const float cosPhi = z;
  const float sinPhi = sqrtf(-(__fsub_rn(__fmul_rn(z, z), 1.0f)));
  const bool small = sinPhi < 1e-5f;
  const float cosTheta = small ? 1.0f : __fdiv_rn(x, sinPhi);
  const float sinTheta = small ? 0.0f : __fdiv_rn(y, sinPhi);
  const float stcp = __fmul_rn(sinTheta, cosPhi), stsp = __fmul_rn(sinTheta, sinPhi);
  float* o = Y + ray * 12;
  o[0] = 1.0f;
  o[1] = cosTheta;
  o[2] = stcp;
  o[3] = stsp;
  o[4] = __fsub_rn(__fmul_rn(cosTheta, cosTheta), (float)(1.0 / 3));
  o[5] = __fmul_rn(stcp, cosTheta);
  o[6] = __fmul_rn(stsp, cosTheta);
  o[7] = __fmul_rn(__fsub_rn(__fmul_rn(__fmul_rn(stcp, cosPhi), 2.0f), sinTheta), sinTheta);
  o[8] = __fmul_rn(__fmul_rn(stsp, sinTheta), cosPhi);
  o[9] = o[10] = o[11] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// L2 loss (Loss.l2Loss("L2Loss",1), runNerf.scala:64): mean over n*3 of (label-pred)^2, and its gradient
// loss_acc[slot] accumulates the SUM in double; d_rgb = 2 (pred - label) / (3n)
// ---------------------------------------------------------------------------------------------
__global__ void l2_loss_kernel(const float* __restrict__ pred, const float* __restrict__ label, int64_t n3,
                               double* __restrict__ loss_acc, float* __restrict__ d_rgb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double sq = 0.0;
  if (i < n3) {
    const float diff = __fsub_rn(label[i], pred[i]);
    sq = (double)__fmul_rn(diff, diff);
    d_rgb[i] = __fmul_rn(-diff, __fdiv_rn(2.0f, (float)n3));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_acc, part[0] + part[1] + part[2] + part[3]);
}

// ---------------------------------------------------------------------------------------------
// Adam, DJL form (Optimizer.adam, runNerf.scala:65): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// w -= lr' m / (sqrt(v) + eps) with lr' = lr sqrt(1-b2^t)/(1-b1^t) computed on the host in double.
// ---------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lrp, float grad_scale) {
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = __fmul_rn(g[i], grad_scale);
    const float mi = __fadd_rn(__fmul_rn(m[i], b1), __fmul_rn(gi, 1.0f - b1));
    const float vi = __fadd_rn(__fmul_rn(v[i], b2), __fmul_rn(__fmul_rn(gi, gi), 1.0f - b2));
    m[i] = mi;
    v[i] = vi;
    p[i] = __fsub_rn(p[i], __fdiv_rn(__fmul_rn(mi, lrp), __fadd_rn(sqrtf(vi), eps)));
  }
}

// XavierInitializer(GAUSSIAN, IN, 2): N(0, 2/fan_in) for matrices, 0 for biases; sd[] per segment.
struct InitSeg { int64_t off, n; float sd; };
__global__ void init_kernel(float* __restrict__ p, InitSeg seg, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < seg.n; i += (int64_t)gridDim.x * blockDim.x)
    p[seg.off + i] = seg.sd > 0.f ? philox_normal(seed, kStreamInit, 0, (uint64_t)(seg.off + i)) * seg.sd : 0.f;
}

// ---------------------------------------------------------------------------------------------
hipError_t launch_pack_chunks(const float* params, const ChunkDesc* descs, int n, float* dst, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_chunks_kernel, dim3(n), dim3(256), 0, s, params, descs, dst);
  return hipGetLastError();
}
hipError_t launch_repack_bf16(const RepackBf16Args& a, hipStream_t s) {
  hipLaunchKernelGGL(repack_bf16_kernel, dim3(a.nfrag + a.nnets * RepackBf16Args::kBvecBlocks), dim3(64), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_pack_frags_f16x3(const float* params, const FragDesc* descs, int n, void* dst, uint32_t* flag, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_frags_f16x3_kernel, dim3(n), dim3(64), 0, s, params, descs, (_Float16*)dst, flag);
  return hipGetLastError();
}
hipError_t launch_pack_bvec(const float* params, const NetLayout& L, int HT, float* bv, float scale, hipStream_t s) {
  hipLaunchKernelGGL(pack_bvec_kernel, dim3(16), dim3(256), 0, s, params, L, HT, bv, scale);
  return hipGetLastError();
}
hipError_t launch_pack_vec(const float* params, const NetLayout& L, int HT, float* vp, hipStream_t s) {
  hipLaunchKernelGGL(pack_vec_kernel, dim3(16), dim3(256), 0, s, params, L, HT, vp);
  return hipGetLastError();
}
hipError_t launch_dir_branch(const float* params, const NetLayout& L, int Rh, const float* viewdir, int64_t n,
                             float* dirvec, float* encT, int64_t Bpad, hipStream_t s) {
  const int rpb = 256 / Rh;
  hipLaunchKernelGGL(dir_branch_kernel, dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, s, params, L, Rh, viewdir,
                     n, dirvec, encT, Bpad);
  return hipGetLastError();
}
hipError_t launch_sh2(const float* viewdir, int64_t n, float* Y, hipStream_t s) {
  hipLaunchKernelGGL(sh2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, viewdir, n, Y);
  return hipGetLastError();
}
hipError_t launch_l2_loss(const float* pred, const float* label, int64_t n3, double* loss_acc, float* d_rgb,
                          hipStream_t s) {
  hipLaunchKernelGGL(l2_loss_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, pred, label, n3, loss_acc,
                     d_rgb);
  return hipGetLastError();
}
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lrp, float grad_scale,
                       hipStream_t s) {
  hipLaunchKernelGGL(adam_kernel, dim3(1024), dim3(256), 0, s, p, g, m, v, n, lrp, grad_scale);
  return hipGetLastError();
}
hipError_t launch_init(float* p, int64_t off, int64_t n, float sd, uint64_t seed, hipStream_t s) {
  InitSeg seg{off, n, sd};
  hipLaunchKernelGGL(init_kernel, dim3(256), dim3(256), 0, s, p, seg, seed);
  return hipGetLastError();
}

}  // namespace nh
