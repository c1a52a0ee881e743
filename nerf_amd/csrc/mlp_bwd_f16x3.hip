// mlp_bwd_f16x3.hip -- backward-data pass of the density/colour MLP with SPLIT-f16 MFMA operands (gfx950): the function
// of mlp_bwd.hip (reverse mode of nerfCoreBlock.forwardInternal, reference nerf/nerfCoreBlock.scala:17-112; the
// reference gets it from libtorch autograd, runNerf.scala:96-105) to fp32 accuracy at matrix-core rates, for training
// steps with dtype f16x3.  It reads what the training forward wrote into the fp32 activation store (ReLU sign bits) and
// writes what mlp_bwd.hip writes (every dY_l^T and the head gradients, fp32, tile-major) for the fp32 weight-gradient
// kernel; machinery and arithmetic of the forward (f16x3_common.hip.h: weights 64 w = hi + lo, three MFMAs per k-step
// into one accumulator).
//
// Gradients need one thing activations do not: a SCALE.  d(loss)/d(pre-activation) of a sample is 1e-4 .. 1e-12, far
// below f16's normal range, and f16 has 5 exponent bits.  Every lane owns one sample (transposed formulation), the
// whole pass is linear in that sample's gradient, so each lane carries its sample's gradient times a power of two
// 2^e chosen so that the largest element sits near 2^5 (exact: powers of two; the stores multiply it out again).
// e is set from the head gradients and re-chosen behind every layer from the largest element that layer produced
// (one layer late: a part's outputs are split while it runs, so it uses the maximum of its INPUT); a single layer
// would have to grow a gradient 900-fold to reach f16's range, and shrinking only moves the small elements of the
// sample towards the 2^-25-of-the-unit floor of the unscaled low halves.  Elements more than 2^7 below the sample's
// largest are exact to 2^-30 of the largest instead of 2^-22 of themselves -- in a weight gradient, a sum over
// samples and features dominated by the large elements, that is below fp32's own rounding.
//
// Backward stream (api.hip: build_bwd_stream, pairs): (mode 1) Wc^T, HT tiles x 1 k-step | head: WT tiles x (2 HT
// k-steps of Wf^T + 1 k-step whose k = 0 is the density row) | layers D-1 .. 1: WT tiles x 2 WT k-steps of W_l^T.
#include "f16x3_common.hip.h"
#include <algorithm>
#include "launchers.h"

namespace nh {
namespace x3 {

template <int NT>
struct GradRegs { Pair v[NT][2]; };   // [k-tile][k-step], a sample's gradient times its current scale

template <int WT>
struct SelGrad {
  const GradRegs<WT>& g;
  template <int KS_IDX> __device__ __forceinline__ const Pair& b() const { return g.v[KS_IDX / 2][KS_IDX % 2]; }
};
template <int HT>
struct SelHead {    // k-steps 0..2HT-1: d(head pre-activation); 2HT: the density k-step (k = 0 is d sigma)
  const GradRegs<HT>& dh;
  const Pair& dsig;
  template <int KS_IDX> __device__ __forceinline__ const Pair& b() const {
    if constexpr (KS_IDX < 2 * HT) return dh.v[KS_IDX / 2][KS_IDX % 2];
    else return dsig;
  }
};
struct SelOne {
  const Pair& x;
  template <int KS_IDX> __device__ __forceinline__ const Pair& b() const { return x; }
};

__device__ __forceinline__ float* uniform_ptr_b(float* p) {   // wave-uniform pointer, made so for the compiler (asm "s")
  const uint64_t u = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (float*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

template <int R_ = 0>
__device__ __forceinline__ void store_tile16(float* dst, uint32_t voff, const float (&v)[16]) {   // one 32-row tile
  if constexpr (R_ < 16) { store_reg<0, R_>(dst, voff, v[R_]); store_tile16<R_ + 1>(dst, voff, v); }
}

// largest value of a wave (DPP: row_shr 1/2/4/8, row_bcast 15/31; valid in lane 63; v >= 0, idle lanes read 0) into
// the workgroup's slot in LDS: non-negative floats order like their bit patterns.  The workgroup publishes its slots
// to device memory ONCE, when it has run out of tiles: a global atomic per wave and part (330 000 per step on ten
// addresses) sat in the in-order vmcnt queue in front of the ring's LDS-DMA and cost 1.25 ms of the kernel's 4.4.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ void publish_max(unsigned* slot, float v, int lane) {
  v = fmaxf(v, dpp_f<0x111, 0xf>(v)); v = fmaxf(v, dpp_f<0x112, 0xf>(v)); v = fmaxf(v, dpp_f<0x114, 0xf>(v));
  v = fmaxf(v, dpp_f<0x118, 0xf>(v)); v = fmaxf(v, dpp_f<0x142, 0xa>(v)); v = fmaxf(v, dpp_f<0x143, 0xc>(v));
  if (lane == 63) atomicMax(slot, __builtin_bit_cast(unsigned, v));
}

// scale exponents stay inside [-kExpLim, kExpLim] so that no multiplier (2^(e_out - e_in - 6), 2^(-e_in - 6)) leaves fp32
constexpr int kExpLim = 56;
constexpr int kTargetLog2 = 5;   // the largest element of a sample's scaled gradient sits in [2^4, 2^5)
__device__ __forceinline__ int floor_log2_bits(float x) { return (int)((__builtin_bit_cast(uint32_t, x) >> 23) & 0xffu) - 127; }
__device__ __forceinline__ int clamp_exp(int e) { return min(max(e, -kExpLim), kExpLim); }
__device__ __forceinline__ float pow2i(int k) { return __builtin_bit_cast(float, (uint32_t)(k + 127) << 23); }   // |k| <= 126

// the same split as split_scaled_into with a per-lane multiplier: x = a * k (exact), hi = f16(x), lo = f16(x - hi)
__device__ __forceinline__ void split_by_into(Pair& p, int i, float a0, float a1, float k) {
  uint32_t h, l;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(a0), "v"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(a1), "v"(k));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(k), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(k), "v"(h));
  p.hi[i] = h;
  p.lo[i] = l;
}

// Epilogue of a part whose output tiles are gradients w.r.t. the pre-activations of a ReLU layer with NT tiles:
// mask by the layer's sign bits, write to the fp32 store (true value), keep -- rescaled -- as the next B operand.
// `a` = 64 * 2^e_in * (W^T g).
template <int NT>
struct EpiGrad {
  GradRegs<NT>& out;
  Ring& rg;
  float* dst;            // wave-uniform: this wave's tile block of the gradient buffer (mlp_common.hip.h: tile_block)
  uint32_t voff;         // tile_voff(lane)
  const uint32_t (&bits)[4];   // the layer's sign-bit words of this lane (mlp_common.hip.h: relu_mask_words)
  float kx;              // 2^(e_out - e_in - 6): accumulator -> next operand
  float kst;             // 2^(-e_in - 6):        accumulator -> true gradient
  float amax;            // running max |masked accumulator| of this lane
  static constexpr int kPieces = 8;
  template <int OT_IDX, int P> __device__ __forceinline__ void piece(const f32x16& a) {
    const uint32_t w = bits[OT_IDX >> 1];
    const float x0 = a[2 * P], x1 = a[2 * P + 1];
    const int m0 = __builtin_amdgcn_sbfe((int)w, 16 * (OT_IDX & 1) + 2 * P, 1);       // 0 | -1
    const int m1 = __builtin_amdgcn_sbfe((int)w, 16 * (OT_IDX & 1) + 2 * P + 1, 1);
    const float v0 = __builtin_bit_cast(float, __builtin_bit_cast(int, x0) & m0);
    const float v1 = __builtin_bit_cast(float, __builtin_bit_cast(int, x1) & m1);
    amax = fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1)));
    store_reg<OT_IDX, 2 * P>(dst, voff, v0 * kst);
    store_reg<OT_IDX, 2 * P + 1>(dst, voff, v1 * kst);
    rg.note(2);
    split_by_into(out.v[OT_IDX][P / 4], P % 4, v0, v1, kx);
  }
};

#ifdef X3B_STAMPS   // diagnostic build only (scripts/x3b_stamps.py): cycle stamps of one wave's 4th tile
__device__ unsigned long long g_x3b_stamps[32];
#define X3B_STAMP(i) do { if (blockIdx.x == 8 && wave == 1 && tile == blockIdx.x + 3 * (int)gridDim.x && lane == 0) g_x3b_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define X3B_STAMP(i)
#endif
template <int WT, int MODE>
__global__ __launch_bounds__(64 * kWaves, 1) void mlp_bwd_f16x3_kernel(NetArgs net, const char* __restrict__ stream,
                                                                       const float* __restrict__ d_sigma,
                                                                       const float* __restrict__ d_craw,
                                                                       const float* __restrict__ dirfeat, int64_t M, int S,
                                                                       int nblk, int ntiles, ActStore st,
                                                                       float* __restrict__ amax_out) {
  constexpr int R = 32 * WT;
  constexpr int HT = MODE == 1 ? (WT >= 2 ? WT / 2 : 1) : 1;
  constexpr int Rh = 32 * HT;
  constexpr int NW = (WT + 1) / 2, NWH = (HT + 1) / 2;
  extern __shared__ __attribute__((aligned(1024))) float smem[];   // ring only
  __shared__ unsigned s_amax[kMaxD + 2];   // largest |gradient| bits per amax slot, over this workgroup's tiles

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = net.D;
  const int64_t Mp = st.Mpad;

  Ring rg;
  rg.stream = stream;
  rg.ring = smem;
  rg.ring_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)smem + (uint32_t)lane * 16u;
  rg.wave = wave;
  rg.lane = lane;
  rg.voff = (uint32_t)(wave * (kBlkBytes / kWaves) + lane * 16);
  rg.nblk = nblk;
  if (tid < kMaxD + 2) s_amax[tid] = 0u;   // visible behind the barrier in start()
  rg.start();
  const uint32_t svoff = tile_voff(lane);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    X3B_STAMP(0);
    const int m = tile * kTileSamples + wave * 32 + li;   // host guarantees M < 2^31
    const bool valid = m < (int)M;
    const int mc = valid ? m : (int)M - 1;
    const int ray = (int)((uint32_t)mc / (uint32_t)S);
    const int tix = tile * kWaves + wave;
    // sign bits of a layer: 4 words per lane (word w at + w * Mpad); slot D = the colour hidden layer
    auto load_bits = [&](uint32_t (&b)[4], int slot, int nw) __attribute__((always_inline)) {
#pragma unroll
      for (int w = 0; w < 4; ++w) b[w] = w < nw ? __builtin_nontemporal_load(mask_ptr(st.mask, slot, h, w, Mp, m)) : 0u;
      rg.note(nw);
    };
    uint32_t bitsA[4], bitsB[4], bitsH[4];
    if constexpr (MODE == 1) load_bits(bitsH, D, NWH);
    load_bits(bitsA, D - 1, NW);
    const float dsg = valid ? d_sigma[m] : 0.f;
    float dc[3] = {0.f, 0.f, 0.f};
    if (valid) { dc[0] = d_craw[(int64_t)m * 3]; dc[1] = d_craw[(int64_t)m * 3 + 1]; dc[2] = d_craw[(int64_t)m * 3 + 2]; }

    // the sample's scale: largest head gradient -> [2^4, 2^5); zero gradients keep scale 1
    const float amax0 = fmaxf(fmaxf(fabsf(dsg), fabsf(dc[0])), fmaxf(fabsf(dc[1]), fabsf(dc[2])));
    const int e0 = amax0 > 0.f ? clamp_exp(kTargetLog2 - 1 - floor_log2_bits(amax0)) : 0;
    const float s0 = pow2i(e0);
    publish_max(s_amax + D + 1, amax0, lane);

    // head gradients go out feature-major behind the hidden-layer rows: d sigma, then d craw^T (mlp_bwd.hip)
    float* const dhb = uniform_ptr_b(tile_block(st.dh, tix, Rh + 32));
    if (h == 0) {
      dhb[(Rh + 0) * 32 + li] = dsg;
      dhb[(Rh + 1) * 32 + li] = dc[0];
      dhb[(Rh + 2) * 32 + li] = dc[1];
      dhb[(Rh + 3) * 32 + li] = dc[2];
    }
    rg.note(4);

    // feature 0 (= element 0 of lane half 0 of k-step 0) = d sigma; features 0..2 = d craw for the Wc^T part
    Pair dsig, dcol;
#pragma unroll
    for (int i = 0; i < 4; ++i) { dsig.hi[i] = 0u; dsig.lo[i] = 0u; dcol.hi[i] = 0u; dcol.lo[i] = 0u; }
    if (h == 0) {
      split_into(dsig, 0, dsg * s0, 0.f);
      split_into(dcol, 0, dc[0] * s0, dc[1] * s0);
      split_into(dcol, 1, dc[2] * s0, 0.f);
    }

    X3B_STAMP(1);
    // ---- d(head pre-activation), scale e0
    GradRegs<HT> dh;
    float amax_prev;   // max |accumulator| of the part just finished (both lane halves), its input scale in e_prev
    int e_prev;
    if constexpr (MODE == 1) {
      // Wc^T . d craw, masked by the colour hidden layer's ReLU; kept at scale e0 (the head part adds the density
      // k-step, whose operand is at e0, into the same accumulators)
      EpiGrad<HT> epi{dh, rg, dhb, svoff, bitsH, pow2i(-6), pow2i(-e0 - 6), 0.f};
      run_part<1, HT, false, true>(rg, 0u, SelOne{dcol}, epi);
      publish_max(s_amax + D, epi.amax * epi.kst, lane);
    } else {
      // mode 2: d q[3j+k] = d c[k] * Y_j(dir) (rows 0..26); mode 0: rows 0..2 = d c
      const float* Y = dirfeat + (int64_t)ray * 12;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tile_row(0, r, h);
        v[r] = 0.f;
        if (MODE == 2) { if (row < 27) v[r] = dc[row % 3] * Y[row / 3]; }
        else if (row < 3) v[r] = dc[row];
      }
      store_tile16(dhb, svoff, v);
      rg.note(16);
      float vm = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) vm = fmaxf(vm, fabsf(v[r]));
      publish_max(s_amax + D, vm, lane);
#pragma unroll
      for (int p = 0; p < 8; ++p) split_into(dh.v[0][p / 4], p % 4, v[2 * p] * s0, v[2 * p + 1] * s0);
    }

    X3B_STAMP(2);
    // ---- head backward: d feat = Wf^T . dh + w_sigma . d sigma, masked by the last trunk layer's ReLU
    GradRegs<WT> g, gn;
    {
      if (D >= 2) load_bits(bitsB, D - 2, NW);
      EpiGrad<WT> epi{g, rg, uniform_ptr_b(tile_block(st.dy + (int64_t)(D - 1) * R * Mp, tix, R)), svoff, bitsA, pow2i(-6),
                      pow2i(-e0 - 6), 0.f};
      run_part<2 * HT + 1, WT, false, true>(rg, 0u, SelHead<HT>{dh, dsig}, epi);
      publish_max(s_amax + D - 1, epi.amax * epi.kst, lane);
      amax_prev = fmaxf(epi.amax, __shfl_xor(epi.amax, 32));
      e_prev = e0;
    }
    X3B_STAMP(3);
    // ---- trunk, last layer first: dY_{l-1} = ReLU'_{l-1} . (W_l^T . dY_l); two layers per trip (g -> gn -> g)
    int e_in = e0;   // scale of the operands the next part consumes
    auto layer = [&](int l, const GradRegs<WT>& in, GradRegs<WT>& out, const uint32_t (&use)[4], uint32_t (&fetch)[4])
                     __attribute__((always_inline)) {
      if (l >= 2) load_bits(fetch, l - 2, NW);
      // scale of this part's OUTPUT operands: from the largest element of its input (amax_prev, taken at scale e_prev)
      const int e_out = amax_prev > 0.f ? clamp_exp(kTargetLog2 - 1 + e_prev + 6 - floor_log2_bits(amax_prev)) : e_in;
      EpiGrad<WT> epi{out, rg, uniform_ptr_b(tile_block(st.dy + (int64_t)(l - 1) * R * Mp, tix, R)), svoff, use,
                      pow2i(e_out - e_in - 6), pow2i(-e_in - 6), 0.f};
      run_part<2 * WT, WT, false, true>(rg, 0u, SelGrad<WT>{in}, epi);
      publish_max(s_amax + l - 1, epi.amax * epi.kst, lane);
      amax_prev = fmaxf(epi.amax, __shfl_xor(epi.amax, 32));
      e_prev = e_in;
      e_in = e_out;
      X3B_STAMP(3 + D - l);
    };
    int l = D - 1;
    for (; l >= 2; l -= 2) {
      layer(l, g, gn, bitsB, bitsA);
      layer(l - 1, gn, g, bitsA, bitsB);
    }
    if (l >= 1) layer(l, g, gn, bitsB, bitsA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
  __syncthreads();
  if (tid < D + 2 && s_amax[tid] != 0u) atomicMax(reinterpret_cast<unsigned*>(amax_out) + tid, s_amax[tid]);
}

template <int WT, int MODE>
static hipError_t launch_bwd_one(const ActStore& st, const NetArgs& net, const void* stream_w, const float* d_sigma,
                                 const float* d_craw, const float* dirfeat, int64_t M, int S, int nblk, int n_cu, float* amax,
                                 hipStream_t stream) {
  const size_t lds = (size_t)kRing * kBlkBytes;
  const int ntiles = (int)((M + kTileSamples - 1) / kTileSamples);
  const dim3 grid((unsigned)std::min(ntiles, n_cu)), block(64 * kWaves);
  hipLaunchKernelGGL((mlp_bwd_f16x3_kernel<WT, MODE>), grid, block, lds, stream, net, (const char*)stream_w, d_sigma, d_craw,
                     dirfeat, M, S, nblk, ntiles, st, amax);
  return hipGetLastError();
}

}  // namespace x3

#define NH_X3B_CASES(F)                                                                                                 \
  switch ((W / 32) * 4 + mode) {                                                                                        \
    case 2 * 4 + 0: F(2, 0); case 2 * 4 + 1: F(2, 1); case 2 * 4 + 2: F(2, 2);                                          \
    case 4 * 4 + 0: F(4, 0); case 4 * 4 + 1: F(4, 1); case 4 * 4 + 2: F(4, 2);                                          \
    case 8 * 4 + 0: F(8, 0); case 8 * 4 + 1: F(8, 1); case 8 * 4 + 2: F(8, 2);                                          \
    default: return hipErrorInvalidValue;                                                                               \
  }

// once per device, with that device current (nerfhip_create)
hipError_t prepare_mlp_bwd_f16x3(int W, int mode) {
#define NH_X3B_PREP(WT_, MODE_) \
  return hipFuncSetAttribute((const void*)x3::mlp_bwd_f16x3_kernel<WT_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, x3::kRing * x3::kBlkBytes)
  NH_X3B_CASES(NH_X3B_PREP)
#undef NH_X3B_PREP
}

#ifdef X3B_STAMPS
extern "C" int nerfhip_debug_x3b_stamps(unsigned long long* out) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(x3::g_x3b_stamps), sizeof(unsigned long long) * 32) != hipSuccess;
}
#endif
hipError_t launch_mlp_bwd_f16x3(const ActStore& st, const NetArgs& net, const void* stream_w, const float* d_sigma,
                                const float* d_craw, const float* dirfeat, int64_t M, int S, int nblk, int n_cu, float* amax,
                                hipStream_t stream) {
  const int W = net.W, mode = net.mode;
#define NH_X3B_LAUNCH(WT_, MODE_) \
  return x3::launch_bwd_one<WT_, MODE_>(st, net, stream_w, d_sigma, d_craw, dirfeat, M, S, nblk, n_cu, amax, stream)
  NH_X3B_CASES(NH_X3B_LAUNCH)
#undef NH_X3B_LAUNCH
}

}  // namespace nh
