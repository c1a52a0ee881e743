// common.h -- shared declarations of libnerfhip (host + device).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nerfhip.h"

namespace nh {

constexpr int kMaxD = NERFHIP_MAX_D;
constexpr int kEncPad = 64;      // positional encoding rows, zero padded (posL <= 10 -> E <= 63)
constexpr int kTile = 128;       // samples per workgroup of the fused MLP kernels (4 waves x 32)
constexpr int kMaxChunks = 192;  // weight chunks of one network's forward stream

// Canonical flat layout of ONE network's parameters (DJL child order).  Offsets in floats.
struct NetLayout {
  int D, W, E, Ed, H2;
  int mode;              // 0: Linear(3) | 1: view-dir head | 2: SH head  (nerfCoreBlock.scala:52-83)
  int head_rows;         // rows of the first head matrix: H2 | 27 | 3
  int fan_in[kMaxD];
  uint32_t skip_mask;    // bit l set: output of layer l is concatenated with the encoding
  int64_t w_off[kMaxD], b_off[kMaxD];
  int64_t ws_off, bs_off;
  int64_t wf_off, bf_off;   // first head matrix [head_rows][W] and its bias
  int64_t wd_off, bd_off;   // mode 1: [H2][Ed]
  int64_t wc_off, bc_off;   // mode 1: [3][H2]
  int64_t count;
};

// One weight chunk of a packed stream: 32 k-values x R rows.
struct ChunkDesc {
  int64_t src_off;     // float offset of the matrix in the canonical network parameters
  int32_t ld;          // row length (fan_in) of the canonical matrix
  int32_t transposed;  // 0: A[i][k] = src[i*ld + c0 + k]   1: A[i][k] = src[(c0 + k)*ld + i]
  int32_t c0;          // first canonical column (or row, when transposed) of this chunk; -1 = all zero
  int32_t kvalid;      // k < kvalid are real, the rest zero padding
  int32_t ivalid;      // i < ivalid are real rows
  int32_t R;           // rows in the chunk (multiple of 32)
  uint32_t dst_off;    // float offset in the packed stream
  uint32_t pad;
};

// One 1-KiB fragment of a packed bf16 weight stream (bf16_common.hip.h): 32 rows x 16 k-values of one matrix.
struct FragDesc {
  int64_t src_off;     // float offset of the matrix in the canonical network parameters; < 0: all-zero fragment
  int32_t ld;          // row length of the canonical matrix
  int32_t row0;        // canonical row of lane i = 0 (lane i -> row0 + i)
  int32_t nrows;       // rows of the canonical matrix (rows >= nrows read as zero)
  int32_t col0;        // canonical column of feature 0 of the 32-feature k-tile
  int32_t kvalid;      // features of the k-tile that exist (the rest read as zero)
  int32_t s;           // k-step inside the k-tile (0 | 1)
  int32_t transposed;  // 1: element (row, col) = src[col*ld + row];  2: not transposed, 16 rows x 32 k-values in the lane order of
                       // the 16x16x32 MFMA's A operand (mlp_fwd_bf16_t16.hip; s unused)
};

// Kernel-side description of one network (passed by value).
struct NetArgs {
  const float* params;         // canonical fp32 parameters of this network
  const float* vecpack;        // 16-byte aligned copies of the small vectors (see pack.hip):
                               //   [l*W] b_l | [D*W] w_sigma | [(D+1)*W] head bias (zero padded to HT*32)
                               //   [(D+2)*W] wc [3][HT*32] zero padded | [(D+5)*W] b_sigma, bc[0..2]
  const void* wpack_fwd;       // packed forward stream (fp32 or bf16)
  const void* wpack_bwd;       // packed backward (transposed) stream
  const uint32_t* off_fwd;     // [n_fwd+1] element offsets of the forward chunks
  const uint32_t* off_bwd;     // [n_bwd+1]
  int32_t n_fwd, n_fwd_trunk, n_bwd;   // n_fwd_trunk: chunks up to and excluding the colour head
  int32_t D, W, E, mode, HT, head_rows, H2;
  uint32_t skip_mask;
};

inline int64_t vecpack_floats(int D, int W) { return (int64_t)(D + 5) * W + 4; }

// Activation store of one training pass over M samples, fp32.  Every buffer is TILE-MAJOR: samples are grouped
// in tiles of 32 (one wave of the fused MLP kernels) and a buffer with `rows` feature rows holds
//     element(row, m) = buf[((m >> 5) * rows + row) * 32 + (m & 31)]
// so the 32-sample x rows block a wave writes -- and the block the weight-gradient GEMM stages per K-chunk -- is
// ONE contiguous rows*128-byte piece of HBM (sequential DRAM pages, a handful of TLB entries) instead of
// `rows` 128-byte pieces megabytes apart.
struct ActStore {
  float* enc;    // [Mpad/32][64][32]
  float* act;    // [D] x [Mpad/32][W][32]   post-ReLU outputs of the trunk layers (layer l at act + l*W*Mpad)
  float* hc;     // [Mpad/32][HT*32][32]  mode 1: post-ReLU colour hidden layer
  float* dy;     // [D] x [Mpad/32][W][32]   gradients w.r.t. the pre-activations (written by backward)
  float* dh;     // [Mpad/32][HT*32 + 32][32] head gradients: rows < HT*32 d(head pre-act), row HT*32 = d sigma,
                 //                    rows HT*32+1..+3 = d craw^T (mode 1)
  uint32_t* mask;  // [(D+1)][2 (lane half)][4 words][Mpad] ReLU sign bits of act (slot D: hc): bit 16*(t&1)+r of
                   // word t/2 <-> register r of tile t; lets the backward pass fetch 16 B instead of 512 B per layer
  int64_t Mpad;
};
// bf16 training store (cfg.dtype == NERFHIP_BF16).  Every buffer is in MFMA-operand layout: per 32-sample tile
// and per 32-feature k-tile kt two 1-KiB fragments (k-step s = 0, 1) of [lane = 32h + sample][8 bf16], element e
// of lane half h holding feature 32kt + bf_feature(s,h,e) (bf16_common.hip.h) -- exactly the registers the fused
// kernels hold, so a wave stores/loads a fragment with ONE 16-byte access per lane, and the bf16 weight-gradient
// kernel turns it into sample-contiguous MFMA operands with ds_read_b64_tr_b16 (wgrad_bf16.hip).
//   fragment (tile, kt, s) of a buffer with KT k-tiles: byte offset ((tile*KT + kt)*2 + s) * 1024
struct BfStore {
  uint16_t* enc;     // KT = 2            positional encoding
  uint16_t* act;     // [D] x (KT = W/32) post-ReLU trunk activations (layer l at act + l*W*Mpad elements)
  uint16_t* hc;      // KT = HT           mode 1: post-ReLU colour hidden layer
  uint16_t* dy;      // [D] x (KT = W/32) gradients w.r.t. the trunk pre-activations
  uint16_t* dh;      // KT = HT + 1       head gradients; tile HT: feature 0 = d sigma, 1..3 = d craw
  uint16_t* encdir;  // KT = 1            mode 1: positionCode(viewdir) of the sample's ray
  uint32_t* relu;    // [D] x 16 B per lane of every 32-sample tile: ReLU' bits of the trunk layers (bf16_common.hip.h:
                     // relu_bits), layer l at relu + l*Mpad*8 words
  int64_t Mpad;
};

__host__ __device__ inline int64_t tiled_index(int64_t row, int64_t m, int64_t rows) {
  return ((m >> 5) * rows + row) * 32 + (m & 31);
}

}  // namespace nh
