// bf16_common.hip.h -- machinery of the bf16-MFMA MLP kernels (gfx950, v_mfma_f32_32x32x16_bf16).
//
// Same transposed formulation as the fp32 kernels (mlp_common.hip.h): activations H^T [features][samples] with
// the sample on the lane, weights as the A operand, fp32 accumulators.  What changes with bf16:
//   * one MFMA contracts 16 features (lane half h holds 8 consecutive-in-register bf16) and takes 8 passes, i.e.
//     16x the fp32 rate: the A fragments (1 KiB per wave per MFMA) can no longer come from L2 -- they are staged
//     through LDS (1 KiB per wave per MFMA = 128 B/clk/CU, half of the LDS read rate);
//   * the loop is OUTPUT-TILE outer: all K of one 32-row output tile is accumulated before the next tile
//     starts, so only 2 x NS accumulator tiles are live, the ReLU + bf16 conversion of tile t runs in the shadow
//     of tile t+1's MFMAs, and the converted tile is directly a B operand of the next layer:
//     accumulator register r = 8s + e of a tile holds feature  F(s,h,e) = 16s + (e&3) + 8(e>>2) + 4h,
//     so registers 8s..8s+7 of both lane halves are exactly the 16 features of k-step s; the weight fragment of
//     (k-tile kt, step s, output tile ot) is packed in that same order (pack.hip: repack_bf16_kernel).
//
// Weight stream: one flat sequence of 1-KiB fragments in consumption order, every layer part padded to a
// multiple of 32 fragments.  It moves through a ring of kRing LDS slots in fixed 32-KiB blocks by LDS-DMA
// (global_load_lds 16 B; every wave moves 8 KiB of each block): during block k every wave passes ONE sync point
// P_k (own pieces of block k+1 landed [counted vmcnt] -> s_barrier -> request block k+3), placed 4 fragments
// before the end of the block so the fragment prefetch (3 ahead, asm ds_read_b128 with counted lgkmcnt) runs
// across block boundaries without a bubble.  The allocation is padded so requests past the end stay in bounds.
#pragma once
#include "mlp_common.hip.h"

namespace nh {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBlkFrags = 32;
constexpr int kBlkBytes = kBlkFrags * 1024;
// Tiling of a kernel: NS sample tiles (of 32) per wave, WAVES waves per workgroup (256 samples either way).
//   <1, 8>  two waves per SIMD (256 registers each): one wave's epilogue VALU work and LDS waits hide behind the other's
//           MFMAs; every wave reads a 1-KiB fragment out of LDS per MFMA -- the training kernels (their stores assume it);
//   <2, 4>  one wave per SIMD (512 registers), every fragment feeds TWO MFMAs (both sample tiles): half the LDS read
//           traffic per MFMA -- the inference forward kernel, whose LDS read port was as busy as its matrix pipe.
template <int NS_, int WAVES_>
struct BfCfg {
  static constexpr int NS = NS_, WAVES = WAVES_;
  static constexpr int TILE = WAVES_ * NS_ * 32;          // samples per workgroup
  static constexpr int DMA = kBlkBytes / WAVES_ / 1024;   // LDS-DMA instructions per wave per block
};
using BfCfg1 = BfCfg<1, 8>;
constexpr int kNS = BfCfg1::NS;                // the training kernels' tiling, by its old names
constexpr int kBfWaves = BfCfg1::WAVES;
constexpr int kBfTile = BfCfg1::TILE;
constexpr int kRing = 4;
#ifndef NH_BF_PD
#define NH_BF_PD 4
#endif
constexpr int kPD = NH_BF_PD;           // fragments fetched ahead
constexpr int kSyncFrag = kBlkFrags - kPD - 1;   // position of the sync point inside a block (before the prefetch crosses into the next)

// feature (inside a 32-feature tile) held by element e of k-step s in lane half h
__device__ __host__ constexpr int bf_feature(int s, int h, int e) { return 16 * s + (e & 3) + 8 * (e >> 2) + 4 * h; }

// s_waitcnt vmcnt(n) for a wave-uniform run-time n >= 4 (rounded down to a supported count: stricter, never weaker)
__device__ __forceinline__ void wait_vm_at_most(int n) {
  if (n >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

template <class C>
struct BfRingT {
  const char* stream;   // packed fragments (global), nblk blocks; consumed cyclically (persistent workgroups start
                        // the next sample tile on block 0 again while the ring keeps running)
  float* ring;          // LDS ring, slot 0
  uint32_t ring_addr;   // LDS byte address of slot 0 + lane*16
  int nblk;             // blocks in the stream
  int blk;              // consumption counter (never wraps; slot = blk % kRing)
  int req, req_src;     // request counter (slot = req % kRing) and the stream block it maps to
  int wave, lane;
  // vmcnt retires in order, so "block blk+1 has landed" = "at most Y operations outstanding", Y = the VMEM
  // operations this wave issued behind that block's pieces: the pieces of block blk+2 plus the activation stores /
  // mask loads of the training kernels, which note() counts (young_a: between the requests of blocks blk+1 and
  // blk+2, young_b: since).  Counting them lets those stay in flight across sync points; an operation that is not
  // noted only makes the wait stricter.  Every noted operation must be exactly >= 1 VMEM instruction.
  int young_a, young_b;
  __device__ __forceinline__ void note(int n) { young_b += n; }
  __device__ __forceinline__ void request() {
    const char* src = stream + (size_t)req_src * kBlkBytes + wave * (kBlkBytes / C::WAVES) + lane * 16;
    char* dst = reinterpret_cast<char*>(ring) + (req & (kRing - 1)) * kBlkBytes + wave * (kBlkBytes / C::WAVES);
#pragma unroll
    for (int p = 0; p < C::DMA; ++p)
      __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(src + p * 1024), NERFHIP_LPTR(dst + p * 1024), 16, 0, 0);
    asm volatile("" ::: "memory");   // no later memory operation may be issued in front of the pieces
    ++req;
    req_src = req_src + 1 == nblk ? 0 : req_src + 1;
  }
  // kernel prologue: blocks 0..2 requested, block 0 visible
  __device__ __forceinline__ void start() {
    blk = 0; req = 0; req_src = 0;
    asm volatile("" ::: "memory");
    request(); request(); request();
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * C::DMA) : "memory");
    __builtin_amdgcn_s_barrier();
    young_a = young_b = 0;
  }
  // sync point P_blk: block blk+1 visible to every wave, slot of block blk-1 handed to block blk+3
  __device__ __forceinline__ void sync_point() {
    wait_vm_at_most(young_a + C::DMA + young_b);
    __builtin_amdgcn_s_barrier();
    request();
    young_a = young_b;
    young_b = 0;
  }
  __device__ __forceinline__ uint32_t slot_addr(int b) const { return ring_addr + (uint32_t)(b & (kRing - 1)) * kBlkBytes; }
};
using BfRing = BfRingT<BfCfg1>;

template <int IMM>
__device__ __forceinline__ void lds_read_frag(bf16x8& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int IMM>
__device__ __forceinline__ void lds_read_f4(f32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
// These waits are NOT optional timing hints.  The compiler considers the destination of an asm ds_read written at the
// statement and re-uses the same physical VGPRs afterwards -- in the inference kernel 56 of the 186 fragment-destination
// registers double as address operands of global loads / stores / LDS-DMA -- so a fragment that lands late overwrites
// whatever lives there by then: garbage operands at best, a memory fault when it is an address (a development build with
// these waits compiled out faulted the GPU, scripts/README.md).  scripts/check_inflight.py proves on the final ISA that no
// instruction reads or writes such a register while its load may be outstanding.
template <int N>
__device__ __forceinline__ void wait_lgkm_n() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N)); }

// accumulator-init rows of one output tile (bias, or any per-feature fp32 vector laid out like a bias):
// 4 x 16 B per lane, features 32 ot + 8q + 4h + {0..3}
struct BiasBuf { f32x4 q[4]; };
template <int OT_>
__device__ __forceinline__ void bias_read(BiasBuf& b, uint32_t addr) {   // addr: vector base + h*16
  lds_read_f4<(32 * OT_ + 0) * 4>(b.q[0], addr);
  lds_read_f4<(32 * OT_ + 8) * 4>(b.q[1], addr);
  lds_read_f4<(32 * OT_ + 16) * 4>(b.q[2], addr);
  lds_read_f4<(32 * OT_ + 24) * 4>(b.q[3], addr);
}
__device__ __forceinline__ f32x16 bias_value(BiasBuf& b) {   // call only behind the wait that retired the reads
  asm volatile("" : "+v"(b.q[0]), "+v"(b.q[1]), "+v"(b.q[2]), "+v"(b.q[3]));
  f32x16 c;
#pragma unroll
  for (int q = 0; q < 4; ++q) { c[4 * q] = b.q[q][0]; c[4 * q + 1] = b.q[q][1]; c[4 * q + 2] = b.q[q][2]; c[4 * q + 3] = b.q[q][3]; }
  return c;
}

// ---- one layer part: OT output tiles x KS k-steps, fragments j = ot*KS + ks ---------------------------------
// BSel:  template <int KS_IDX> const bf16x8& b(int n) const      B operand of k-step KS_IDX for sample tile n
// Epi:   static constexpr int kPieces;  template <int OT_IDX, int P> void piece(f32x16 (&acc)[NS])
//        the epilogue of a finished output tile, cut into kPieces pieces which are spread over the k-steps of the
//        NEXT tile (accumulators are double buffered), so its VALU work issues between that tile's MFMAs instead
//        of stalling on the last MFMA's result with the matrix pipe idle; the last tile's pieces run at the end.
template <int KS, int OT, bool BIAS = true>
struct PartSched {
  static constexpr int NF = KS * OT;
  static constexpr int kBiasStep = 1;   // k-step of tile ot at which the accumulator-init rows of tile ot+1 are requested
                                        // (behind the MFMA of step 0 that consumed the single buffer)
  // LDS operations issued behind the newest operation step j depends on (its fragment; at the first k-step of a
  // tile also that tile's init rows), up to the wait of step j -- by replaying the issue order:
  //   prime: init rows of tile 0 (4 reads), fragments 0..min(kPD,NF)-1;  step t: fragment t+kPD, then (t % KS ==
  //   kBiasStep) init rows of the next tile
  static constexpr int pending(int j) {
    int pos = BIAS ? 4 : 0, need = (j == 0) ? pos : 0;
    for (int f = 0; f < (NF < kPD ? NF : kPD); ++f) { pos += 1; if (f == j) need = pos; }
    for (int t = 0; t <= j; ++t) {
      if (t + kPD < NF) { pos += 1; if (t + kPD == j && pos > need) need = pos; }
      if (BIAS && t % KS == kBiasStep && t / KS + 1 < OT) { pos += 4; if ((t / KS + 1) * KS == j && pos > need) need = pos; }
    }
    return pos - need;
  }
  // is the sync point of the block that fragment j lies in executed right after step j?
  static constexpr bool sync_after(int j) { return (j % kBlkFrags == kSyncFrag) || (j == NF - 1 && (j % kBlkFrags) < kSyncFrag); }
};

template <int KS, int NP>
constexpr int piece_offset(int p) { return (p * KS) / NP + (KS >= 8 ? 1 : 0); }
template <int NS, int KS, class Epi, int OT_IDX, int OFF, int P = 0>
struct EpiPieces {   // runs the pieces of tile OT_IDX scheduled at k-step offset OFF (OFF < 0: all)
  static __device__ __forceinline__ void run(Epi& epi, f32x16 (&acc)[NS]) {
    if constexpr (P < Epi::kPieces) {
      if constexpr (OFF < 0 || piece_offset<KS, Epi::kPieces>(P) == OFF) epi.template piece<OT_IDX, P>(acc);
      EpiPieces<NS, KS, Epi, OT_IDX, OFF, P + 1>::run(epi, acc);
    }
  }
};

template <class C, int KS, int OT, class BSel, class Epi, bool BIAS, int J>
struct PartStep {
  using S = PartSched<KS, OT, BIAS>;
  static constexpr int kNS = C::NS;
  static __device__ __forceinline__ void run(BfRingT<C>& rg, int blk0, uint32_t bias_addr, bf16x8 (&fr)[kPD + 1], BiasBuf& bias,
                                             f32x16 (&acc)[2][C::NS], const BSel& bsel, Epi& epi) {
    constexpr int ot = J / KS, ks = J % KS;
    if constexpr (J + kPD < S::NF) {
      constexpr int jj = J + kPD;
      lds_read_frag<(jj % kBlkFrags) * 1024>(fr[jj % (kPD + 1)], rg.slot_addr(blk0 + jj / kBlkFrags));
    }
    if constexpr (BIAS && ks == S::kBiasStep && ot + 1 < OT) bias_read<ot + 1>(bias, bias_addr);
    wait_lgkm_n<S::pending(J)>();
    asm volatile("" : "+v"(fr[J % (kPD + 1)]));
    if constexpr (ks == 0) {
      f32x16 c0;
      if constexpr (BIAS) c0 = bias_value(bias);
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = 0.f;
      }
#pragma unroll
      for (int n = 0; n < kNS; ++n)
        acc[ot & 1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[J % (kPD + 1)], bsel.template b<ks>(n), c0, 0, 0, 0);
    } else {
#pragma unroll
      for (int n = 0; n < kNS; ++n)
        acc[ot & 1][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[J % (kPD + 1)], bsel.template b<ks>(n), acc[ot & 1][n], 0, 0, 0);
    }
    if constexpr (ot >= 1) EpiPieces<C::NS, KS, Epi, ot - 1, ks>::run(epi, acc[(ot - 1) & 1]);
    if constexpr (ot == OT - 1 && ks == KS - 1) EpiPieces<C::NS, KS, Epi, ot, -1>::run(epi, acc[ot & 1]);
    if constexpr (S::sync_after(J)) { rg.blk = blk0 + J / kBlkFrags; rg.sync_point(); }
    if constexpr (J + 1 < S::NF) PartStep<C, KS, OT, BSel, Epi, BIAS, J + 1>::run(rg, blk0, bias_addr, fr, bias, acc, bsel, epi);
  }
};

template <int NF, int F>
__device__ __forceinline__ void prime_frags_bf(bf16x8 (&fr)[kPD + 1], uint32_t addr) {   // fragments 0..min(kPD,NF)-1
  if constexpr (F < kPD && F < NF) {
    lds_read_frag<F * 1024>(fr[F], addr);
    prime_frags_bf<NF, F + 1>(fr, addr);
  }
}
// Runs one part that starts at block rg.blk (block-aligned) and leaves rg.blk at the next part's first block.
// bias_addr: LDS byte address of the part's accumulator-init vector + h*16.
template <int KS, int OT, bool BIAS = true, class C, class BSel, class Epi>
__device__ __forceinline__ void run_part(BfRingT<C>& rg, uint32_t bias_addr, const BSel& bsel, Epi& epi) {
  static_assert(!BIAS || KS >= 2 || OT == 1, "the init rows of the next tile are requested at k-step 1");
  static_assert(OT == 1 || piece_offset<KS, Epi::kPieces>(Epi::kPieces - 1) < KS, "epilogue pieces must fit into the next tile");
  using S = PartSched<KS, OT, BIAS>;
  const int blk0 = rg.blk;
  bf16x8 fr[kPD + 1];
  BiasBuf bias;
  f32x16 acc[2][C::NS];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (BIAS) bias_read<0>(bias, bias_addr);
  prime_frags_bf<S::NF, 0>(fr, rg.slot_addr(blk0));
  PartStep<C, KS, OT, BSel, Epi, BIAS, 0>::run(rg, blk0, bias_addr, fr, bias, acc, bsel, epi);
  wait_lgkm_n<0>();
  __builtin_amdgcn_sched_barrier(0);
  rg.blk = blk0 + (S::NF + kBlkFrags - 1) / kBlkFrags;
}

// round to bf16 (RNE), then ReLU on the packed pairs: a negative bf16 has its sign bit set, i.e. is a negative
// int16, so one v_pk_max_i16 against 0 clears both halves (-0 -> +0 as well) -- 1 VALU per 2 values instead of 2-4.
// accumulator tile -> the two k-step B operands of the next layer
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 relu_bf16x8(bf16x8 x) {
  i16x8 v = __builtin_bit_cast(i16x8, x);
  const i16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  v = __builtin_elementwise_max(v, zero);
  return __builtin_bit_cast(bf16x8, v);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// registers 8 s .. 8 s + 7 of an accumulator tile -> B operand of k-step s of the next layer:
// per pair one v_cvt_pk_bf16_f32 (RNE) and one v_pk_max_i16
template <int S_>
__device__ __forceinline__ bf16x8 relu_pack_half(const f32x16& a) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 v = {a[8 * S_ + 2 * i], a[8 * S_ + 2 * i + 1]};
    const bf16x2 b = __builtin_convertvector(v, bf16x2);
    const i16x2 zero = {0, 0};
    r[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, b), zero));
  }
  return __builtin_bit_cast(bf16x8, r);
}
// ReLU' bits of a trunk layer, 16 B per lane (lane = sample li, half h, as everywhere): word ot>>1, bit
//   8 (ot & 1) + 4 s + i  (+16 for the odd element)   <->  output tile ot, k-step s, elements 2i / 2i+1
// of the fragment, i.e. feature 32 ot + bf_feature(s, h, 2i [+1]).  The forward kernel gathers them from the packed
// post-ReLU pairs (non-zero <=> positive), the backward kernel expands them into 16-bit AND masks.
template <int OT_IDX, int S_>
__device__ __forceinline__ void relu_bits_gather(u32x4& words, const bf16x8& act) {
  typedef unsigned short u16x2_ __attribute__((ext_vector_type(2)));
  const u32x4 r = __builtin_bit_cast(u32x4, act);
  const u16x2_ one = {1, 1};
  unsigned d = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned x = r[i];   // (a bit_cast of the vector ELEMENT expression r[i] reads element 0 with this clang)
    d |= __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2_, x), one)) << i;
  }
  words[OT_IDX >> 1] |= d << (8 * (OT_IDX & 1) + 4 * S_);
}
template <int OT_IDX, int S_>
__device__ __forceinline__ u32x4 relu_bits_expand(const u32x4& words) {
  u32x4 m;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned t = (words[OT_IDX >> 1] >> (8 * (OT_IDX & 1) + 4 * S_ + i)) & 0x00010001u;
    m[i] = t * 0xffffu;                  // 0/1 per half -> 0x0000/0xffff per half
  }
  return m;
}
__device__ __forceinline__ void relu_pack(const f32x16& a, bf16x8& s0, bf16x8& s1) {
  s0 = relu_pack_half<0>(a);
  s1 = relu_pack_half<1>(a);
}

// positionCode value for the bf16 path: hardware sin/cos (v_sin_f32 takes revolutions, |arg| <= 256, so the
// argument is reduced with fract first).  Absolute error <= ~3e-4 at the largest frequency (|x 2^9| ~ 3000 rad:
// fp32 rounding of x/2pi), an order of magnitude below the bf16 rounding (4e-3 relative) applied right after.
__device__ __forceinline__ float code_value_fast(float px, float py, float pz, int e, int E) {
  if (e >= E) return 0.f;
  if (e < 3) return e == 0 ? px : (e == 1 ? py : pz);
  const int t = e - 3;
  const int f = t / 6;
  const int rem = t - 6 * f;
  const int c = rem >= 3 ? rem - 3 : rem;
  const float x = c == 0 ? px : (c == 1 ? py : pz);
  const float rev = __builtin_amdgcn_fractf(x * (float)(1 << f) * 0.15915494309189535f);
  return rem >= 3 ? __builtin_amdgcn_cosf(rev) : __builtin_amdgcn_sinf(rev);
}

}  // namespace nh
