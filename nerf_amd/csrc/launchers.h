// launchers.h -- host-callable launch functions of the HIP kernels (defined in the .hip files).
#pragma once
#include "common.h"

namespace nh {

// ray_kernels.hip
hipError_t launch_stratified(const float* bounds, const float* t_rand, int64_t n, int N, int lin_disp, int perturb,
                             uint64_t seed, uint32_t step, float* z, hipStream_t s);
hipError_t launch_composite_fwd(const float* sigma, const float* noise, const float* craw, const float* z, int64_t n,
                                int S, float noise_std, int white, uint64_t seed, uint32_t step, uint32_t stream,
                                float* w, float* rgb, float* depth, float* acc, hipStream_t s);
hipError_t launch_composite_bwd(const float* sigma, const float* noise, const float* craw, const float* z,
                                const float* d_rgb, int64_t n, int S, float noise_std, int white, uint64_t seed,
                                uint32_t step, uint32_t stream, float* d_sigma, float* d_craw, hipStream_t s);
hipError_t launch_sample_pdf(const float* w, const float* z, const float* u_rand, int64_t n, int N, int NI, int perturb,
                             uint64_t seed, uint32_t step, float* cdf, int32_t* ind, float* zs, float* z_fine,
                             hipStream_t s);
hipError_t launch_gen_rays(int H, int Wd, float focal, const float* c2w, float near, float far, int use_ndc, int row0,
                           int nrows, float* rays_o, float* rays_d, float* bounds, hipStream_t s);
hipError_t launch_to_u8(const float* rgb, int64_t n, uint8_t* out, hipStream_t s);
hipError_t launch_random(uint64_t seed, uint32_t stream, uint32_t step, int64_t n, int normal, float* out, hipStream_t s);
size_t composite_bwd_lds_bytes(int S);
size_t sample_pdf_lds_bytes(int N, int NI, int perturb);

// pack.hip
hipError_t launch_pack_chunks(const float* params, const ChunkDesc* descs, int n, float* dst, hipStream_t s);
hipError_t launch_pack_vec(const float* params, const NetLayout& L, int HT, float* vp, hipStream_t s);
hipError_t launch_dir_branch(const float* params, const NetLayout& L, int Rh, const float* viewdir, int64_t n,
                             float* dirvec, float* encT, int64_t Bpad, hipStream_t s);
hipError_t launch_sh2(const float* viewdir, int64_t n, float* Y, hipStream_t s);
hipError_t launch_l2_loss(const float* pred, const float* label, int64_t n3, double* loss_acc, float* d_rgb,
                          hipStream_t s);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lrp, float grad_scale,
                       hipStream_t s);
hipError_t launch_init(float* p, int64_t off, int64_t n, float sd, uint64_t seed, hipStream_t s);

// mlp_fwd.hip
hipError_t launch_mlp_fwd_f32(bool train, const NetArgs& net, const float* rays_o, const float* rays_d, const float* z,
                              const float* dirfeat, int64_t M, int S, float* sigma, float* craw, int want_rgb,
                              const ActStore& st, hipStream_t stream);

// mlp_bwd.hip
hipError_t launch_mlp_bwd_f32(const NetArgs& net, const float* d_sigma, const float* d_craw, const float* dirfeat,
                              int64_t M, int S, int has_rgb, const ActStore& st, hipStream_t stream);

// bf16 path
// scale: factor on every entry (1; 64 for the split-f16 kernel, whose accumulators carry the weights' factor)
hipError_t launch_pack_bvec(const float* params, const NetLayout& L, int HT, float* bv, float scale, hipStream_t s);
// all bf16 streams and vectors of up to two networks in one launch (pack.hip: repack_bf16_kernel)
struct RepackBf16Args {
  static constexpr int kMaxSeg = 10, kBvecBlocks = 16;
  struct Seg { const FragDesc* descs; void* dst; const float* params; int32_t first, n; } seg[kMaxSeg];
  int32_t nseg, nfrag, nnets;
  int32_t HT[2];
  const float* vparams[2];
  float* bv[2];
  NetLayout L[2];
};
hipError_t launch_repack_bf16(const RepackBf16Args& a, hipStream_t s);
hipError_t launch_mlp_fwd_bf16(const BfStore* st, const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                               const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                               float* craw, int want_rgb, int nblk, int n_cu, hipStream_t stream);

hipError_t launch_mlp_bwd_bf16(const BfStore& st, const NetArgs& net, const void* stream_w, const float* d_sigma,
                               const float* d_craw, const float* dirfeat, int64_t M, int S, int nblk, int n_cu,
                               hipStream_t stream);
// Kernels that use more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised on EVERY
// device they run on: nerfhip_create calls these with the handle's device current (one process may hold handles on
// several devices, as the reference does: nerfConfig.device: Array[Device], runNerf.scala:67).
hipError_t prepare_wgrad_f32();
hipError_t prepare_wgrad_bf16();
hipError_t prepare_mlp_fwd_bf16(int W, int mode);
// inference forward on v_mfma_f32_16x16x32_bf16 (mlp_fwd_bf16_t16.hip); stream_w in FragDesc::transposed == 2 order
hipError_t prepare_mlp_fwd_bf16_t16(int W, int mode);
hipError_t launch_mlp_fwd_bf16_t16(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                                   const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                                   float* craw, int want_rgb, int nblk, int n_cu, hipStream_t stream);
hipError_t prepare_mlp_bwd_bf16(int W, int mode);
// split-f16 render path (mlp_fwd_f16x3.hip, pack.hip)
hipError_t launch_pack_frags_f16x3(const float* params, const FragDesc* descs, int n, void* dst, uint32_t* flag, hipStream_t s);
hipError_t prepare_mlp_fwd_f16x3(int W, int mode);
// inference forward on v_mfma_f32_16x16x32_f16 (mlp_fwd_f16x3_t16.hip); stream_w: pairs in FragDesc::transposed == 2 order
hipError_t prepare_mlp_fwd_f16x3_t16(int W, int mode);
hipError_t launch_mlp_fwd_f16x3_t16(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                                    const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                                    float* craw, int want_rgb, int nblk, int n_cu, uint32_t* nonfinite_flag, hipStream_t stream);
hipError_t launch_mlp_fwd_f16x3(const NetArgs& net, const void* stream_w, const float* bvec, const float* rays_o,
                                const float* rays_d, const float* z, const float* dirfeat, int64_t M, int S, float* sigma,
                                float* craw, int want_rgb, int nblk, int n_cu, const ActStore* st, uint32_t* nonfinite_flag,
                                hipStream_t stream);
// mlp_bwd_f16x3.hip: backward-data pass of a training step with dtype f16x3 (reads/writes the fp32 activation store)
hipError_t prepare_mlp_bwd_f16x3(int W, int mode);
// amax: [D + 2] floats of this network, zeroed by the caller: slot l = largest |dY_l| written, slot D = largest
// |d(head pre-activation)|, slot D + 1 = largest |d sigma|, |d craw| (for the weight-gradient kernel's operand scale)
hipError_t launch_mlp_bwd_f16x3(const ActStore& st, const NetArgs& net, const void* stream_w, const float* d_sigma,
                                const float* d_craw, const float* dirfeat, int64_t M, int S, int nblk, int n_cu,
                                float* amax, hipStream_t stream);
// wgrad_bf16.hip: one slice of the sample axis of one product, operands in the bf16 store's operand layout
struct WgradBfItem {
  const uint16_t* A;   // fragment (tile 0, first k-tile of the A rows, s = 0)
  const uint16_t* B;
  float* dW;
  float* db;
  int64_t a_stride, b_stride;   // bytes per 32-sample tile of the buffers A and B live in
  int32_t a_rows, b_rows;       // rows (features) of the operands, counted from their first k-tile
  int32_t a_row0;               // rows of A below a_row0 are not part of the product (dW row = row - a_row0)
  int32_t split;                // rows >= split go to dW2 / db2 (row - split) instead: two weight tensors fed by one
  float* dW2;                   // pass over the operands (head matrix + density row); split >= a_rows: unused
  float* db2;
  int32_t ldo, col0;
  int32_t c0, c1;               // slice of the sample axis in 32-sample tiles
  int32_t wb_n, shape;          // as WgradItem
};
hipError_t launch_wgrad_bf16(const WgradBfItem* d_items, const int32_t* d_first, int ngroups, hipStream_t s);
hipError_t launch_expand_encdir_bf16(const float* viewdir, int64_t M, int64_t Mpad, int S, int Ed, uint16_t* out, hipStream_t s);

// wgrad.hip
struct WgradItem {
  const float* A;
  const float* B;
  float* dW;
  float* db;
  int32_t a_stride, b_stride;   // rows per 32-sample tile of the (tile-major, common.h) buffers A and B live in
  int32_t a_rows, b_rows;
  int32_t ldo, col0;
  int32_t c0, c1;      // slice of the sample axis in 32-sample chunks
  int32_t wb_n;        // waves along B (1, 2 or 4); waves along A = 4 / wb_n
  int32_t shape;       // tiles per wave: AT * 16 + BT, each in {1, 2, 4}
  // wgrad_f16x3.hip only: largest |element| of the buffer A lives in (device, written by mlp_bwd_f16x3.hip) and
  // log2 of how many such elements an A value may be the sum of (the per-ray sums of the direction branch)
  const float* amax;
  int32_t amax_shift;
  // wgrad.hip and wgrad_f16x3.hip: an extra VECTOR product that rides on this entry's pass over B -- row x_row of the buffer A
  // lives in (e.g. d sigma behind the head gradient rows) against every B row: dWx[col0 + j] += sum_m e[m] B[j][m],
  // dbx += sum_m e[m]; fp32 FMAs on the B fragments the waves hold anyway.  x_row < 0: none
  int32_t x_row;
  float* dWx;
  float* dbx;
};
hipError_t prepare_wgrad_f16x3();
hipError_t launch_wgrad_f16x3(const WgradItem* d_items, const int32_t* d_first, int ngroups, hipStream_t s);
hipError_t launch_wgrad_f32(const WgradItem* d_items, const int32_t* d_first, int ngroups, hipStream_t s);
hipError_t launch_ray_sum(const float* dh, int dh_rows, int rows, int64_t n, int S, float* out, int out_rows,
                          hipStream_t s);

}  // namespace nh
