/*
 * nerf_oracle.h -- CPU restatement of the reference NeRF hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for libnerfhip.  It restates, in plain fp32 C++ loops, the arithmetic of
 * the reference package `nerf` (reference paths are relative to /root/reference/src/main/scala):
 *   nerf/nerfBlock.scala      (sampling, compositing, inverse-CDF resampling, orchestration)
 *   nerf/nerfCoreBlock.scala  (positional encoding, MLP trunk, density / colour heads, SH2)
 *   nerf/runNerf.scala:63-73,94-106,181-204 (L2 loss, Adam + factor tracker, uint8 render rows)
 *   nerf/getDataSet.scala:12-21,102-112 and nerf/llff.scala:96-132 (ray generation, bounds, NDC)
 *
 * PARITY UNPINNED: the reference has no tests, fixtures or golden vectors, and its arithmetic lives in
 * third-party DJL 0.16.0 + libtorch 1.8.1 (build.sbt:10-13) which cannot run in this environment
 * (no JVM, no network).  The oracle is therefore pinned against (a) closed-form known-answer tests and
 * (b) golden vectors produced by tests/golden/make_golden.py, an independent op-by-op transliteration
 * of the Scala source onto torch CPU ops (the same libtorch op families DJL dispatches to).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (nerf_amd/, libnerfhip.so) never links, imports or calls it.
 *
 * Layout: everything is RAY-MAJOR ([ray][sample]) and row-major; the reference stores [sample][ray]
 * (nerfBlock.scala:33-34) -- the arithmetic per ray is identical.
 */
#ifndef NERF_ORACLE_H
#define NERF_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SKIPS 8
#define ORC_MAX_D 16

/* Mirrors the arithmetic-relevant fields of nerf/nerfConfig.scala:20-43. */
typedef struct orc_config {
  int32_t use_dir;          /* useDir */
  int32_t use_sh;           /* useSH */
  int32_t use_hierarchical; /* useHierarchical */
  int32_t pos_l;            /* posL */
  int32_t dir_l;            /* dirL */
  int32_t D;                /* number of trunk layers (the reference comments swap D/W; code uses D as depth) */
  int32_t W;                /* trunk width */
  int32_t n_skips;
  int32_t skips[ORC_MAX_SKIPS];
  int32_t n_samples;        /* NSamples */
  int32_t n_importance;     /* NImportance */
  float raw_noise_std;      /* rawNoiseStd */
  int32_t white_bkgd;       /* whiteBkgd */
  int32_t lin_disp;         /* linDisp */
  int32_t perturb;          /* perturb */
  float lrate;              /* lrate */
  int32_t lrate_decay;      /* lrateDecay (in thousands of steps) */
} orc_config;

/* Optional tensors of one forward pass; any pointer may be NULL. */
typedef struct orc_io {
  /* random inputs (NULL => deterministic path / zero noise) */
  const float* t_rand;   /* [B][N]        U(0,1) for givePerturb (nerfBlock.scala:55) */
  const float* u_rand;   /* [B][NI]       U(0,1), unsorted, for givePerturbPDF (nerfBlock.scala:71) */
  const float* noise_c;  /* [B][N]        N(0,1) for coarse raw density (last column unused) */
  const float* noise_f;  /* [B][N+NI]     N(0,1) for fine raw density   (last column unused) */
  /* outputs */
  float* rgb_fine;    /* [B][3] */
  float* rgb_coarse;  /* [B][3] (training && hierarchical only) */
  float* depth;       /* [B]  sum w*z of the last (fine) pass, nerf_origin/nerf_model.scala:218 */
  float* acc;         /* [B]  sum w */
  float* z_coarse;    /* [B][N] */
  float* w_coarse;    /* [B][N] */
  float* cdf;         /* [B][N-1] */
  int32_t* ind;       /* [B][NI] */
  float* z_samples;   /* [B][NI]  before the merge-sort */
  float* z_fine;      /* [B][S] S = N+NI (or N when not hierarchical) */
  float* w_fine;      /* [B][S] */
  float* sigma_c;     /* [B][N]  raw density, last column 0 */
  float* sigma_f;     /* [B][S] */
  float* craw_c;      /* [B][N][3] raw colour (training only for coarse) */
  float* craw_f;      /* [B][S][3] */
  /* test hook (last, so that the layout above is unchanged): when non-NULL and hierarchical, these fine depths replace the
   * resampler's output for the rest of the pass (fine network, compositing, loss, backward).  samplePdf carries no gradient
   * (stopGradient, nerfBlock.scala:81), so the pass from the fine depths on is a function of (params, z_fine) alone: a test
   * can hand in the DEVICE's fine depths and compare everything behind the sampler without the sampler's `den < 1e-5`
   * discontinuity in the way (the sampler itself is compared bit for bit on identical inputs elsewhere). */
  const float* z_fine_in;   /* [B][S] */
} orc_io;

int64_t orc_net_param_count(const orc_config* c);   /* one network */
int64_t orc_param_count(const orc_config* c);       /* coarse (if hierarchical) + fine */
int32_t orc_pos_width(const orc_config* c);         /* 3*(1+2*posL) */

void orc_linspace(int32_t n, float* out);
void orc_position_code(const float* x, int64_t n, int32_t L, float* out);   /* [n][3] -> [n][3+6L] */
void orc_sh2(const float* dir, int64_t n, float* out);                      /* [n][3] -> [n][9] */
void orc_stratified(const orc_config* c, int64_t B, const float* bounds, const float* t_rand, float* z);
void orc_composite(const orc_config* c, int64_t B, int32_t S, const float* sigma, const float* noise,
                   const float* craw, const float* z, float* w, float* rgb, float* depth, float* acc);
void orc_composite_bwd(const orc_config* c, int64_t B, int32_t S, const float* sigma, const float* noise,
                       const float* craw, const float* z, const float* w, const float* d_rgb,
                       float* d_sigma, float* d_craw);
void orc_sample_pdf(const orc_config* c, int64_t B, const float* w, const float* z, const float* u_rand,
                    float* cdf, int32_t* ind, float* z_samples, float* z_fine);
/* One network on [B][S] sample positions z along rays; sigma [B][S] (last col 0), craw [B][S][3] or NULL. */
void orc_mlp_forward(const orc_config* c, const float* net_params, int32_t is_coarse, int32_t training,
                     int64_t B, int32_t S, const float* rays_o, const float* rays_d, const float* viewdir,
                     const float* z, float* sigma, float* craw);
/* The same network with every dot product summed in the order the fp32 MFMA kernels use (nerf_amd/csrc/mlp_common.hip.h:
 * k = 32kt + 8q + 4h + j, fmaf chain): restates the KERNEL's rounding, not the reference's, so that tests can demand
 * bit-for-bit agreement of the MLP stage where no sin/cos is involved (posL = dirL = 0) and ~1e-6 elsewhere. */
void orc_mlp_forward_korder(const orc_config* c, const float* net_params, int32_t want_rgb, int64_t B, int32_t S,
                            const float* rays_o, const float* rays_d, const float* viewdir, const float* z,
                            float* sigma, float* craw);
/* Full forward (nerfBlock.forwardInternal). */
void orc_forward(const orc_config* c, const float* params, int32_t training, int64_t B,
                 const float* rays_o, const float* rays_d, const float* bounds, const float* viewdir,
                 orc_io* io);
/* forward(training=1) + L2 loss (runNerf.scala:72-73) + backward; grads has orc_param_count floats. */
void orc_train_grads(const orc_config* c, const float* params, int64_t B,
                     const float* rays_o, const float* rays_d, const float* bounds, const float* viewdir,
                     const float* label, orc_io* io, float* loss_fine, float* loss_coarse, float* grads);
/* DJL Adam + factor tracker (runNerf.scala:65); t = update count starting at 1. */
float orc_lr(const orc_config* c, int64_t t);
void orc_adam(const orc_config* c, int64_t n, float* p, const float* g, float* m, float* v, int64_t t,
              float grad_scale);
void orc_init_params(const orc_config* c, uint64_t seed, float* params);
/* getRaysNp + normalisation + bounds (getDataSet.scala:12-21,110-112), NDC (llff.scala:96-132). */
void orc_gen_rays(int32_t H, int32_t Wd, float focal, const float* c2w /*[3][4]*/, float near, float far,
                  int32_t use_ndc, float* rays_o, float* rays_d, float* bounds);
void orc_to_u8(const float* rgb, int64_t n, uint8_t* out);   /* mul(255).toType(UINT8) runNerf.scala:194 */
int32_t orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
