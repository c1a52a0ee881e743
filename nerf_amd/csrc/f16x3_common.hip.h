// f16x3_common.hip.h -- machinery of the SPLIT-f16 MLP forward kernel (gfx950, v_mfma_f32_32x32x16_f16): fp32-grade
// results at matrix-core rates.
//
// Every fp32 operand is split into two f16 halves of 11 significant bits each and a product of two such numbers is
// taken as three MFMA terms in ONE fp32 accumulator:
//     weights      64 w = w_hi + w_lo ,  w_hi = f16(64 w) ,  w_lo = f16(64 w - w_hi)     (packed once, pack.hip)
//     activations     x = x_hi + x_lo ,  x_hi = f16(x) ,     x_lo = f16(x - x_hi)        (every layer, in registers)
//     acc += w_hi . x_hi + w_hi . x_lo + w_lo . x_hi          acc = 64 (W x + b)   (the bias enters as 64 b)
// (v - f16(v) is exact in fp32, so each half pair is exact up to the rounding of its low half.)  Why the weights carry
// a factor 64: a low half is an UNSCALED residual, 2^-11 of its high half, and f16 has only 5 exponent bits -- below
// |v| = 0.25 the residual is a subnormal f16 and the pair is exact to 2^-25 absolute instead of 2^-22 relative.  For
// the activations that does not matter (the error is multiplied by a weight of O(0.1)); for the weights, N(0, 2/fan_in)
// and multiplied by activations of O(1..10), it would, so they are stored 64 times larger: exact pairs down to
// |w| = 0.004, range |w| < 1023.  The factor is taken out again where the accumulator is read -- inside the
// v_fma_mix{lo,hi}_f16 that make the next layer's halves (f16(acc/64) and f16(acc/64 - hi): two instructions per
// value, no separate conversion, scaling or merge).  The dropped term w_lo . x_lo is 2^-22 of the product.  Emulated
// on a 256 x 256 layer (products exact, one fp32 rounding of the accumulator per MFMA) the result is closer to the
// exact product than a sequential fp32 fmaf chain is (rms 1.5e-7 against 2.0e-7), for weights 10 x smaller or 5 x
// larger alike -- while the three MFMAs run at the f16 rate: 2.5 PF / 3 = 833 TF of fp32-grade contraction against
// the 157 TF of v_mfma_f32_32x32x2_f32.  Range: |x| < 65504 and |w| < 1023.
//
// Structure: that of the bf16 kernels (bf16_common.hip.h) -- transposed formulation, output-tile-outer loop, weight
// fragments through a 4 x 32 KiB LDS ring fed by LDS-DMA, asm ds_read_b128 with counted lgkmcnt, the epilogue of
// tile t cut into pieces that issue between the MFMAs of tile t+1 -- with these differences: the stream carries a
// (hi, lo) PAIR of 1-KiB fragments per (output tile, k-step); a step is one pair and three MFMAs; activations are
// kept as (hi, lo) operand pairs, which needs the 512-register file: ONE wave per SIMD, 4 waves (128 samples) per
// workgroup.
#pragma once
#include "mlp_common.hip.h"
#ifndef NH_X3_PD
#define NH_X3_PD 2
#endif

namespace nh {
namespace x3 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kWScale = 64.0f, kWInv = 1.0f / 64.0f;   // factor on the packed weights and biases (pack.hip)
constexpr int kWaves = 4;              // one per SIMD
constexpr int kTileSamples = kWaves * 32;
constexpr int kBlkFrags = 32;          // 1-KiB fragments per ring block = 16 (hi, lo) pairs
constexpr int kBlkPairs = kBlkFrags / 2;
constexpr int kBlkBytes = kBlkFrags * 1024;
constexpr int kRing = 4;
constexpr int kPD = NH_X3_PD;                 // pairs fetched ahead
constexpr int kSyncPair = kBlkPairs - kPD - 1;
constexpr int kDmaPieces = kBlkBytes / kWaves / 1024;   // LDS-DMA instructions per wave per block (8)

// feature (inside a 32-feature tile) held by element e of k-step s in lane half h: as the bf16 MFMA of the same shape
__device__ __host__ constexpr int feature(int s, int h, int e) { return 16 * s + (e & 3) + 8 * (e >> 2) + 4 * h; }

__device__ __forceinline__ void wait_vm_at_most(int n) {   // s_waitcnt vmcnt(<= n), n >= 8, wave-uniform
  if (n >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

struct Ring {
  const char* stream;   // packed fragment pairs (global), nblk blocks, consumed cyclically
  float* ring;
  uint32_t ring_addr;   // LDS byte address of slot 0 + lane*16
  int nblk, blk, req, req_src, wave, lane;
  // training kernel: VMEM operations of this wave behind the pieces of block blk+1 (young_a: up to the request of
  // block blk+2, young_b: since) -- activation / mask stores, counted by note(); vmcnt retires in order, so counting
  // them lets the stores stay in flight across sync points (as bf16_common.hip.h: BfRing)
  int young_a, young_b;
  __device__ __forceinline__ void note(int n) { young_b += n; }
  uint32_t voff;        // this lane's byte offset inside a block: wave quarter + lane * 16
  __device__ __forceinline__ void request() {
    // one wave moves its quarter of the block with 8 LDS-DMA instructions of 1 KiB; the piece offset rides in the
    // instruction's immediate (applied to the global AND the LDS address, 4 KiB range), so a request costs two
    // address/M0 set-ups instead of eight
    const char* src = stream + (size_t)req_src * kBlkBytes + voff;
    char* dst = reinterpret_cast<char*>(ring) + (req & (kRing - 1)) * kBlkBytes + wave * (kBlkBytes / kWaves);
#pragma unroll
    for (int g = 0; g < kDmaPieces / 4; ++g) {
      __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(src + g * 4096), NERFHIP_LPTR(dst + g * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(src + g * 4096), NERFHIP_LPTR(dst + g * 4096), 16, 1024, 0);
      __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(src + g * 4096), NERFHIP_LPTR(dst + g * 4096), 16, 2048, 0);
      __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(src + g * 4096), NERFHIP_LPTR(dst + g * 4096), 16, 3072, 0);
    }
    asm volatile("" ::: "memory");
    ++req;
    req_src = req_src + 1 == nblk ? 0 : req_src + 1;
  }
  __device__ __forceinline__ void start() {   // blocks 0..2 requested, block 0 visible
    blk = 0; req = 0; req_src = 0;
    asm volatile("" ::: "memory");
    request(); request(); request();
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * kDmaPieces) : "memory");
    __builtin_amdgcn_s_barrier();
    young_a = young_b = 0;
  }
  // sync point P_blk: block blk+1 visible to every wave, slot of block blk-1 handed to block blk+3.  vmcnt retires in
  // order: at most the pieces of block blk+2 may still be outstanding (the kernel's own stores only make it stricter)
  template <bool COUNT>
  __device__ __forceinline__ void sync_point() {
    if constexpr (COUNT) wait_vm_at_most(young_a + kDmaPieces + young_b);
    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(kDmaPieces) : "memory");
    __builtin_amdgcn_s_barrier();
    request();
    if constexpr (COUNT) { young_a = young_b; young_b = 0; }
  }
  __device__ __forceinline__ uint32_t slot_addr(int b) const { return ring_addr + (uint32_t)(b & (kRing - 1)) * kBlkBytes; }
};

template <int IMM>
__device__ __forceinline__ void lds_read_frag(f16x8& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int IMM>
__device__ __forceinline__ void lds_read_f4(f32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int N>
__device__ __forceinline__ void wait_lgkm_n() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N)); }

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

// a (hi, lo) operand pair of one k-step: 8 f16 each, kept as 4 dwords of two
struct Pair { u32x4 hi, lo; };
__device__ __forceinline__ f16x8 as_f16x8(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }
// split two fp32 values into element pair i (0..3) of an operand pair: hi = f16(v), lo = f16(v - hi)
__device__ __forceinline__ void split_into(Pair& p, int i, float v0, float v1) {
  const f32x2 v = {v0, v1};
  const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  uint32_t l;
  // f16(f32(h.half) * -1 + v): the residual straight from the packed halves, one rounding
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(v1));
  p.hi[i] = h;
  p.lo[i] = l;
}
// the same for two accumulator values that still carry the weights' factor: x = a / 64 (exact), hi = f16(x),
// lo = f16(x - hi), each half one fused multiply-add that rounds once to f16
__device__ __forceinline__ void split_scaled_into(Pair& p, int i, float a0, float a1) {
  uint32_t h, l;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(a0), "s"(kWInv));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(a1), "s"(kWInv));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "s"(kWInv), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "s"(kWInv), "v"(h));
  p.hi[i] = h;
  p.lo[i] = l;
}

// ---- one layer part: OT output tiles x KS k-steps; step j = ot*KS + ks consumes fragment pair j ----------------
// BSel:  template <int KS_IDX> const Pair& b() const            B operand pair of k-step KS_IDX
// Epi:   static constexpr int kPieces;  template <int OT_IDX, int P> void piece(const f32x16& acc)   (acc = 64 x the tile)
//        the epilogue of a finished output tile, cut into kPieces pieces spread over the k-steps of the NEXT tile
template <int KS, int OT, bool BIAS = true>
struct PartSched {
  static constexpr int NF = KS * OT;
  static constexpr int kBiasStep = 1;
  // LDS operations issued behind the newest operation step j depends on, up to the wait of step j (replay of the
  // issue order: prime = init rows of tile 0 (4 reads) + pairs 0..min(kPD,NF)-1 (2 reads each); step t: pair t+kPD,
  // then (t % KS == kBiasStep) the init rows of the next tile)
  static constexpr int pending(int j) {
    int pos = BIAS ? 4 : 0, need = (j == 0) ? pos : 0;
    for (int f = 0; f < (NF < kPD ? NF : kPD); ++f) { pos += 2; if (f == j) need = pos; }
    for (int t = 0; t <= j; ++t) {
      if (t + kPD < NF) { pos += 2; if (t + kPD == j && pos > need) need = pos; }
      if (BIAS && t % KS == kBiasStep && t / KS + 1 < OT) { pos += 4; if ((t / KS + 1) * KS == j && pos > need) need = pos; }
    }
    return pos - need;
  }
  static constexpr bool sync_after(int j) { return (j % kBlkPairs == kSyncPair) || (j == NF - 1 && (j % kBlkPairs) < kSyncPair); }
};

template <int KS, int NP>
constexpr int piece_offset(int p) { return (p * KS) / NP; }
template <int KS, class Epi, int OT_IDX, int OFF, int P = 0>
struct EpiPieces {   // runs the pieces of tile OT_IDX scheduled at k-step offset OFF (OFF < 0: all)
  static __device__ __forceinline__ void run(Epi& epi, const f32x16& a) {
    if constexpr (P < Epi::kPieces) {
      if constexpr (OFF < 0 || piece_offset<KS, Epi::kPieces>(P) == OFF) epi.template piece<OT_IDX, P>(a);
      EpiPieces<KS, Epi, OT_IDX, OFF, P + 1>::run(epi, a);
    }
  }
};

// COUNT: the kernel issues VMEM operations of its own between the sync points (Ring::note)
template <int KS, int OT, class BSel, class Epi, bool BIAS, bool COUNT, int J>
struct PartStep {
  using S = PartSched<KS, OT, BIAS>;
  static __device__ __forceinline__ void run(Ring& rg, int blk0, uint32_t bias_addr, f16x8 (&fr)[kPD + 1][2], BiasBuf& bias,
                                             f32x16 (&acc)[2], const BSel& bsel, Epi& epi) {
    constexpr int ot = J / KS, ks = J % KS;
    if constexpr (J + kPD < S::NF) {
      constexpr int jj = J + kPD;
      lds_read_frag<((2 * jj) % kBlkFrags) * 1024>(fr[jj % (kPD + 1)][0], rg.slot_addr(blk0 + jj / kBlkPairs));
      lds_read_frag<((2 * jj + 1) % kBlkFrags) * 1024>(fr[jj % (kPD + 1)][1], rg.slot_addr(blk0 + jj / kBlkPairs));
    }
    if constexpr (BIAS && ks == S::kBiasStep && ot + 1 < OT) bias_read<ot + 1>(bias, bias_addr);
    wait_lgkm_n<(S::pending(J) < 15 ? S::pending(J) : 15)>();   // lgkmcnt is a 4-bit counter: a smaller count is only stricter
    asm volatile("" : "+v"(fr[J % (kPD + 1)][0]), "+v"(fr[J % (kPD + 1)][1]));
    const f16x8& wh = fr[J % (kPD + 1)][0];
    const f16x8& wl = fr[J % (kPD + 1)][1];
    const Pair& b = bsel.template b<ks>();
    if constexpr (ks == 0) {
      f32x16 c0;
      if constexpr (BIAS) c0 = bias_value(bias);
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = 0.f;
      }
      acc[ot & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, as_f16x8(b.hi), c0, 0, 0, 0);
    } else {
      acc[ot & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, as_f16x8(b.hi), acc[ot & 1], 0, 0, 0);
    }
    acc[ot & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, as_f16x8(b.lo), acc[ot & 1], 0, 0, 0);
    acc[ot & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, as_f16x8(b.hi), acc[ot & 1], 0, 0, 0);
    if constexpr (ot >= 1) EpiPieces<KS, Epi, ot - 1, ks>::run(epi, acc[(ot - 1) & 1]);
    if constexpr (ot == OT - 1 && ks == KS - 1) EpiPieces<KS, Epi, ot, -1>::run(epi, acc[ot & 1]);
    if constexpr (S::sync_after(J)) { rg.blk = blk0 + J / kBlkPairs; rg.template sync_point<COUNT>(); }
    if constexpr (J + 1 < S::NF) PartStep<KS, OT, BSel, Epi, BIAS, COUNT, J + 1>::run(rg, blk0, bias_addr, fr, bias, acc, bsel, epi);
  }
};

template <int NF, int F>
__device__ __forceinline__ void prime_pairs(f16x8 (&fr)[kPD + 1][2], uint32_t addr) {   // pairs 0..min(kPD,NF)-1
  if constexpr (F < kPD && F < NF) {
    lds_read_frag<(2 * F) * 1024>(fr[F][0], addr);
    lds_read_frag<(2 * F + 1) * 1024>(fr[F][1], addr);
    prime_pairs<NF, F + 1>(fr, addr);
  }
}
// Runs one part that starts at block rg.blk (block-aligned) and leaves rg.blk at the next part's first block.
template <int KS, int OT, bool BIAS = true, bool COUNT = false, class BSel, class Epi>
__device__ __forceinline__ void run_part(Ring& rg, uint32_t bias_addr, const BSel& bsel, Epi& epi) {
  static_assert(!BIAS || KS >= 2 || OT == 1, "the init rows of the next tile are requested at k-step 1");
  static_assert(kPD < kBlkPairs, "the prefetch must stay inside the block after the current one");
  using S = PartSched<KS, OT, BIAS>;
  const int blk0 = rg.blk;
  f16x8 fr[kPD + 1][2];
  BiasBuf bias;
  f32x16 acc[2];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (BIAS) bias_read<0>(bias, bias_addr);
  prime_pairs<S::NF, 0>(fr, rg.slot_addr(blk0));
  PartStep<KS, OT, BSel, Epi, BIAS, COUNT, 0>::run(rg, blk0, bias_addr, fr, bias, acc, bsel, epi);
  wait_lgkm_n<0>();
  __builtin_amdgcn_sched_barrier(0);
  rg.blk = blk0 + (S::NF + kBlkPairs - 1) / kBlkPairs;
}

}  // namespace x3
}  // namespace nh
