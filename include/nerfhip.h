/*
 * nerfhip.h -- C ABI of libnerfhip.so, the MI355X-native (gfx950, HIP) NeRF render-and-train hot path.
 *
 * Drop-in boundary: this library replaces what sits behind the reference's DJL block `nerfBlock`
 * (reference paths relative to /root/reference/src/main/scala):
 *     nerf/nerfBlock.scala:146-157   forwardInternal(NDList(raysO,raysD,bounds,viewDir), training)
 *     nerf/runNerf.scala:100-106     trainer.forward / calculateLoss / collector.backward / trainer.step
 *     nerf/runNerf.scala:181-204     renderToImage (row-wise evaluate -> *255 -> uint8)
 * The Scala host keeps ray batching, configuration and checkpoints; it reaches these entry points
 * through JNI/Panama.  No torch / Python types appear here: plain pointers, sizes,
 * int error codes.  All tensors are row-major and RAY-major: rays_o/rays_d/viewdir [n][3],
 * bounds [n][2] = (near*|d|, far*|d|) (getDataSet.scala:112), rgb [n][3].
 *
 * Every function returns 0 on success or a negative NERFHIP_E_* code; nerfhip_last_error() gives text.
 * A handle is not thread-safe; distinct handles are independent.  Entry points suffixed _dev take
 * DEVICE pointers (already resident in HBM) and enqueue on the handle's stream without synchronising;
 * the un-suffixed forms take HOST pointers, copy in/out and synchronise.
 */
#ifndef NERFHIP_H
#define NERFHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERFHIP_MAX_SKIPS 8
#define NERFHIP_MAX_D 16

enum {
  NERFHIP_OK = 0,
  NERFHIP_E_INVALID = -1,     /* bad argument / unsupported configuration */
  NERFHIP_E_HIP = -2,         /* a HIP runtime call failed */
  NERFHIP_E_NOMEM = -3,
  NERFHIP_E_STATE = -4,       /* call sequence error (e.g. optimizer_step before backward) */
  NERFHIP_E_COMM = -5         /* RCCL could not be loaded or a collective failed */
};

/* MFMA operand type of the MLP kernels.  F32: v_mfma_f32_32x32x2_f32 everywhere (the parity mode).  BF16: bf16 operands,
 * fp32 accumulation (renders within 1e-2 abs).  F16X3: split-f16 operands (every operand = hi + lo, two f16 halves; three
 * f16 MFMAs per product, fp32 accumulation; weights must stay below 1023 in magnitude) in the render AND the training
 * kernels: fp32-grade results -- held to the same parity tests and tolerances as F32 -- at about 3x the F32 render rate
 * and 2.3x the F32 training rate; parameters, gradients and optimizer state stay fp32.
 * F16X3 range (enforced, never silent): the operands are f16 pairs of 64*w and of the activations, so a weight with
 * |w| >= 1023.5 (or non-finite) and an activation beyond 65504 have no representation.  nerfhip_set_params rejects such
 * weights with NERFHIP_E_INVALID and names the tensor; weights that LEAVE the range in an optimizer step, and activations
 * that overflow in a forward pass (the network's outputs turn non-finite), raise a device flag that the next synchronising
 * call on the handle -- nerfhip_synchronize, get_params, the host-pointer render / train entry points, a _dev training
 * call that returns its loss -- reports as NERFHIP_E_INVALID (and clears).  The remedy is dtype F32. */
enum { NERFHIP_F32 = 0, NERFHIP_BF16 = 1, NERFHIP_F16X3 = 2 };

/* Mirrors the arithmetic-relevant fields of nerf/nerfConfig.scala:20-43 (same meaning, same order),
 * followed by build-side extensions. */
typedef struct nerfhip_config {
  int32_t use_dir;           /* useDir          nerfConfig.scala:20 */
  int32_t use_sh;            /* useSH           :21 */
  int32_t use_hierarchical;  /* useHierarchical :22 */
  int32_t pos_l;             /* posL            :24  (<= 10 in this build) */
  int32_t dir_l;             /* dirL            :25  (<= 4) */
  int32_t D;                 /* trunk depth     :27  (1..16) */
  int32_t W;                 /* trunk width     :28  (32, 64, 128 or 256) */
  int32_t n_skips;
  int32_t skips[NERFHIP_MAX_SKIPS]; /* skips    :29  (each in [0, D-2]) */
  int32_t n_samples;         /* NSamples        :31 */
  int32_t n_importance;      /* NImportance     :32 */
  float raw_noise_std;       /* rawNoiseStd     :34 */
  int32_t white_bkgd;        /* whiteBkgd       :35 */
  int32_t lin_disp;          /* linDisp         :36 */
  int32_t perturb;           /* perturb         :37 */
  float lrate;               /* lrate           :42 */
  int32_t lrate_decay;       /* lrateDecay      :43 */
  /* ---- build extensions ---- */
  int32_t device;            /* HIP device ordinal (nerfConfig.device, :7) */
  int32_t dtype;             /* NERFHIP_F32 | NERFHIP_BF16 | NERFHIP_F16X3 : MFMA operand type of the MLP (F16X3: split-f16
                              * render and training kernels at fp32 accuracy, weights |w| < 1023) */
  uint64_t seed;             /* Philox key for perturb / raw-noise / init */
  int32_t max_rays;          /* largest n any call will pass (workspace is sized once from it) */
  int32_t reserved;
} nerfhip_config;

typedef struct nerfhip nerfhip_t;

/* Optional per-call tensors (host or device pointers, matching the entry point used); NULL = unused.
 * Explicit random inputs make the stochastic branches (givePerturb nerfBlock.scala:50-57, givePerturbPDF
 * :71, addNoise :21) reproducible for parity tests; when NULL and the option is on, a counter-based
 * Philox4x32-10 stream keyed by (seed, step, ray, sample) is used. */
typedef struct nerfhip_aux {
  const float* t_rand;   /* [n][N]     U(0,1) */
  const float* u_rand;   /* [n][NI]    U(0,1), unsorted */
  const float* noise_c;  /* [n][N]     N(0,1), last column ignored */
  const float* noise_f;  /* [n][S]     N(0,1), S = N+NI (or N), last column ignored */
  /* debug outputs */
  float* z_coarse;   /* [n][N] */
  float* w_coarse;   /* [n][N] */
  float* z_fine;     /* [n][S] */
  float* w_fine;     /* [n][S] */
  float* sigma_c;    /* [n][N]   raw density, last column unspecified */
  float* sigma_f;    /* [n][S] */
  float* craw_c;     /* [n][N][3] (training only) */
  float* craw_f;     /* [n][S][3] */
  float* depth;      /* [n]  sum w*z  (nerf_origin/nerf_model.scala:218) */
  float* acc;        /* [n]  sum w    (nerf_origin/nerf_model.scala:219) */
} nerfhip_aux;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* replaces: new nerfBlock(config) + trainer.initialize (nerfBlock.scala:11-19,131-144; runNerf.scala:60-70) */
int nerfhip_create(const nerfhip_config* cfg, nerfhip_t** out);
void nerfhip_destroy(nerfhip_t* h);
const char* nerfhip_last_error(const nerfhip_t* h);   /* h may be NULL: last create() error */
/* Run on a caller-owned HIP stream (hipStream_t as void*); NULL = the library's own NON-BLOCKING stream (the
 * legacy default stream cannot be selected: it is the NULL handle).  The _dev entry points and
 * nerfhip_optimizer_step are asynchronous: a caller that interleaves them with its own GPU work or with
 * collectives on the gradient buffer must either share one explicit stream with the library or call
 * nerfhip_synchronize in between. */
int nerfhip_set_stream(nerfhip_t* h, void* hip_stream);
int nerfhip_synchronize(nerfhip_t* h);

/* ---- parameters: flat fp32, DJL child order (nerfBlock.scala:16-19, nerfCoreBlock.scala:48,85):
 * per network [coarse first when hierarchical]: trunk L0..L(D-1) as (W[out][in], b), density head (W,b),
 * colour head: view-dir (Wf,bf,Wd,bd,W3,b3) | SH (W27,b27) | plain (W3,b3).
 * replaces: block.saveParameters / parameter access (runNerf.scala:128). */
int64_t nerfhip_param_count(const nerfhip_t* h);
int nerfhip_set_params(nerfhip_t* h, const float* src, int64_t n);
int nerfhip_get_params(nerfhip_t* h, float* dst, int64_t n);
/* XavierInitializer(GAUSSIAN, IN, 2) = N(0, 2/fan_in), biases 0 (DJL default). */
int nerfhip_init_params(nerfhip_t* h, uint64_t seed);
/* Adam moments + update count, for true resume (extension; the reference only saves weights). */
int nerfhip_get_opt_state(nerfhip_t* h, float* m, float* v, int64_t n, int64_t* step);
int nerfhip_set_opt_state(nerfhip_t* h, const float* m, const float* v, int64_t n, int64_t step);

/* ---- render: trainer.evaluate(NDList(raysO,raysD,bounds,viewDir)).get(0) (runNerf.scala:192-194),
 * i.e. nerfBlock.forwardInternal with training=false.  rgb [n][3]; aux may be NULL. */
int nerfhip_render(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                   const float* viewdir, int64_t n, float* rgb, nerfhip_aux* aux);
int nerfhip_render_dev(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                       const float* viewdir, int64_t n, float* rgb, nerfhip_aux* aux);
/* nerfhip_render_dev for callers that CANNOT ALLOCATE (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same inputs, same rgb, bit for bit; the per-call
 * intermediates (depths, raw densities, weights, raw colours, direction features) live in a 64-MiB array that is part
 * of the library's device image (one copy per GPU, held until the process ends) instead of in workspace sized from
 * max_rays, so n is NOT bounded by max_rays but by nerfhip_render_static_max_rays(h) (64 MiB / bytes per ray of this
 * configuration; 11 358 rays for the full 64+128 network); larger n -> NERFHIP_E_INVALID.  Calls no hipMalloc / hipFree /
 * event creation (profiling is off inside it).  Every call on a GPU -- from any handle -- uses that GPU's one array:
 * calls that overlap on one GPU race; order them on one stream.  Asynchronous like the other _dev entry points. */
int nerfhip_render_dev_static(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                              const float* viewdir, int64_t n, float* rgb);
int64_t nerfhip_render_static_max_rays(const nerfhip_t* h);
/* + .mul(255).toType(UINT8) (runNerf.scala:194); out [n][3] uint8, clamped to [0,255] then truncated. */
int nerfhip_render_u8(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                      const float* viewdir, int64_t n, uint8_t* out);

/* One whole image from its camera pose: renderToImage (runNerf.scala:181-204) with the rays made on the device --
 * getRaysNp + normalise + bounds (+ ndc) (getDataSet.scala:12-21,105-112; llff.scala:96-132) -> forward(eval) ->
 * .mul(255).toType(UINT8).  c2w host [3][4]; out host [H][W][3] uint8, pixel (row j, col i) at j*W+i.  Any H*W:
 * rendered in blocks of max_rays.  Replaces the H per-row evaluate calls and the upload of H*W*32 bytes of rays. */
int nerfhip_render_pose_u8(nerfhip_t* h, int32_t H, int32_t W, float focal, const float* c2w_host, float near,
                           float far, int32_t ndc, uint8_t* out);
/* Rows [row0, row1) of that image: renderToImage's row loop (runNerf.scala:187-199: one evaluate per image ROW, rows
 * stacked in order) cut into contiguous row blocks, one block per GPU -- BASELINE configs[2], "rays sharded across 8 GPUs".
 * out host [row1-row0][W][3]; the blocks of all ranks, stacked in rank order, ARE nerfhip_render_pose_u8's image bit for
 * bit (rays are independent; there is no collective on this path).  row0 == row1 renders nothing and succeeds. */
int nerfhip_render_pose_rows_u8(nerfhip_t* h, int32_t H, int32_t W, float focal, const float* c2w_host, float near,
                                float far, int32_t ndc, int32_t row0, int32_t row1, uint8_t* out);

/* ---- training ------------------------------------------------------------------------------- */
/* trainer.forward(inputs) in training mode (runNerf.scala:100): fine and coarse rgb, [n][3] each
 * (rgb_coarse may be NULL; it is only produced when hierarchical). */
int nerfhip_forward_train(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                          const float* viewdir, int64_t n, float* rgb_fine, float* rgb_coarse,
                          nerfhip_aux* aux);
/* forward(training) + L2 loss (runNerf.scala:64,72-73) + backward (:103): leaves d(loss)/d(params) in
 * the handle's flat gradient buffer.  loss_out[0] = fine MSE, loss_out[1] = coarse MSE (host floats,
 * may be NULL: then nothing is copied back and the call does not synchronise). */
int nerfhip_train_backward(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                           const float* viewdir, const float* label, int64_t n, float* loss_out,
                           nerfhip_aux* aux);
int nerfhip_train_backward_dev(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                               const float* viewdir, const float* label, int64_t n, float* loss_out,
                               nerfhip_aux* aux);
/* Device address of the flat fp32 gradient buffer (nerfhip_param_count floats, parameter order): the
 * data-parallel host all-reduces THIS buffer (one RCCL call) between train_backward and optimizer_step
 * (replaces DJL's per-parameter reduce inside trainer.step, runNerf.scala:95-106). */
int nerfhip_grad_buffer(nerfhip_t* h, float** dev_ptr, int64_t* n);
int nerfhip_get_grads(nerfhip_t* h, float* dst, int64_t n);
/* Batch.split(devices, false) (runNerf.scala:95) can leave a device without rays when the last batch of an epoch is
 * smaller than the device count: that replica calls this instead of train_backward -- zero gradient, ready for the
 * all-reduce and optimizer_step -- so that every rank still takes part in the step's collective. */
int nerfhip_zero_grads(nerfhip_t* h);
/* trainer.step() (runNerf.scala:106): Adam(b1 .9, b2 .999, eps 1e-7), lr = lrate * f^t with
 * f = float(0.1^(1/(lrateDecay*1000))) (runNerf.scala:65).  grad_scale multiplies the gradient first
 * (1/world_size for mean-over-devices data parallelism, 1 otherwise). */
int nerfhip_optimizer_step(nerfhip_t* h, float grad_scale);
/* train_backward + optimizer_step(1) in one call: one single-device training iteration. */
int nerfhip_train_step(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                       const float* viewdir, const float* label, int64_t n, float* loss_out);
int nerfhip_train_step_dev(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                           const float* viewdir, const float* label, int64_t n, float* loss_out);

/* ---- data-parallel training: the exchange step, native RCCL (no torch, no Python) -------------
 * replaces: the multi-device branch of the reference's train step -- nerfConfig.device: Array[Device]
 * (nerfConfig.scala:7), .optDevices(config.device) (runNerf.scala:67), next.split(trainer.getDevices, false) (:95),
 * per-shard forward/backward (:97-104) and trainer.step() (:106), inside which DJL's parameter server adds the
 * per-device gradients parameter by parameter through device 0.  Here: ONE in-place ncclAllReduce(sum) of the flat
 * gradient buffer on RCCL (xGMI inside a node) on the handle's stream, then optimizer_step(1/world) = mean over
 * devices, so g devices x B rays equal one device running g*B rays.  One handle = one replica on one device.
 * librccl.so.1 is bound at run time on the first nerfhip_dp_* call (NERFHIP_E_COMM if it cannot be loaded). */
#define NERFHIP_DP_ID_BYTES 128
/* rank 0 makes the rendezvous id (ncclGetUniqueId); the HOST carries its 128 bytes to the other ranks */
int nerfhip_dp_unique_id(uint8_t* id_out /* [NERFHIP_DP_ID_BYTES] */);
/* one process (or thread) per device: joins the communicator; blocks until all `world` ranks have called */
int nerfhip_dp_init(nerfhip_t* h, int32_t rank, int32_t world, const uint8_t* id /* [NERFHIP_DP_ID_BYTES] */);
/* one host thread owning all devices (the reference's shape, runNerf.scala:95-104): handle i becomes rank i of n */
int nerfhip_dp_init_all(nerfhip_t** handles, int32_t n);
int nerfhip_dp_world(const nerfhip_t* h, int32_t* rank, int32_t* world);
/* parameters + Adam moments + update count of `root` to every rank (DJL copies the initialised parameters to every
 * device in trainer.initialize, runNerf.scala:70).  Collective; synchronises. */
int nerfhip_dp_broadcast_state(nerfhip_t* h, int32_t root);
/* in-place all-reduce(sum) of d(loss)/d(params) between train_backward and optimizer_step; asynchronous */
int nerfhip_dp_allreduce_grads(nerfhip_t* h);
int nerfhip_dp_allreduce_grads_all(nerfhip_t** handles, int32_t n);   /* single-thread form (grouped) */
/* train_backward_dev + dp_allreduce_grads + optimizer_step(1/world): one data-parallel iteration of this rank */
int nerfhip_dp_train_step_dev(nerfhip_t* h, const float* rays_o, const float* rays_d, const float* bounds,
                              const float* viewdir, const float* label, int64_t n, float* loss_out);
int nerfhip_dp_destroy(nerfhip_t* h);   /* also done by nerfhip_destroy */

/* ---- single kernels on DEVICE pointers (parity tests and roofline micro-benchmarks) ---------- */
/* getSample + givePerturb (nerfBlock.scala:47-57,63-64): bounds [n][2] -> z [n][N]. */
int nerfhip_k_stratified(nerfhip_t* h, const float* bounds, const float* t_rand, int64_t n, float* z);
/* getWeight + weighted sum + addBkgd (nerfBlock.scala:24-45,115): sigma,z [n][S], craw [n][S][3]
 * (NULL: weights only) -> w [n][S], rgb [n][3], depth [n], acc [n] (each may be NULL). */
int nerfhip_k_composite_fwd(nerfhip_t* h, const float* sigma, const float* noise, const float* craw,
                            const float* z, int64_t n, int32_t S, float* w, float* rgb, float* depth,
                            float* acc);
/* reverse mode of the above: d_rgb [n][3] -> d_sigma [n][S] (last column 0), d_craw [n][S][3]. */
int nerfhip_k_composite_bwd(nerfhip_t* h, const float* sigma, const float* noise, const float* craw,
                            const float* z, const float* d_rgb, int64_t n, int32_t S, float* d_sigma,
                            float* d_craw);
/* samplePdf (nerfBlock.scala:74-106): w,z [n][N] -> cdf [n][N-1], ind [n][NI], z_samples [n][NI],
 * z_fine [n][N+NI] (cdf/ind/z_samples may be NULL). */
int nerfhip_k_sample_pdf(nerfhip_t* h, const float* w, const float* z, const float* u_rand, int64_t n,
                         float* cdf, int32_t* ind, float* z_samples, float* z_fine);
/* nerfCoreBlock.forwardInternal (nerfCoreBlock.scala:107-112) of network `net` (0 = coarse when
 * hierarchical, last = fine) on sample depths z [n][S]: sigma [n][S], craw [n][S][3] (may be NULL). */
int nerfhip_k_mlp_forward(nerfhip_t* h, int32_t net, const float* rays_o, const float* rays_d,
                          const float* viewdir, const float* z, int64_t n, int32_t S, float* sigma,
                          float* craw);
/* getRaysNp + normalise + bounds (+ndc) on device (getDataSet.scala:12-21,105-112; llff.scala:96-132):
 * c2w host [3][4]; outputs device [H*W][3],[H*W][3],[H*W][2], pixel (row j, col i) at index j*W+i. */
int nerfhip_k_gen_rays(nerfhip_t* h, int32_t H, int32_t W, float focal, const float* c2w_host, float near,
                       float far, int32_t ndc, float* rays_o, float* rays_d, float* bounds);

/* The library's own random streams (used when nerfhip_aux leaves t_rand / u_rand / noise_* NULL and the option is on;
 * the reference draws manager.randomUniform / randomNormal there, nerfBlock.scala:21,55,71).  Counter-based
 * Philox4x32-10: value = f(seed, stream, step, index), index = flat index of the tensor the stream replaces
 * (t_rand [n][N], u_rand [n][NI], noise_c [n][N], noise_f [n][S]; NERFHIP_RNG_INIT: flat parameter index, step 0, seed =
 * the seed given to nerfhip_init_params).  `step` counts the forward passes of the handle: nerfhip_rng_step gives the
 * step the LAST forward used and the one the next will use.  k_random writes n values of a stream to DEVICE memory
 * (uniform in (0,1) for PERTURB / PDF, N(0,1) otherwise), so a test can replay a stochastic call with explicit tensors. */
enum { NERFHIP_RNG_PERTURB = 1, NERFHIP_RNG_PDF = 2, NERFHIP_RNG_NOISE_COARSE = 3, NERFHIP_RNG_NOISE_FINE = 4, NERFHIP_RNG_INIT = 5 };
int nerfhip_k_random(nerfhip_t* h, int32_t stream, uint64_t seed, uint32_t step, int64_t n, float* out);
int nerfhip_rng_step(const nerfhip_t* h, uint32_t* last, uint32_t* next);

/* Device time (ms) of the kernels of the last render/train call, by class, measured with HIP events
 * on the handle's stream when profiling is enabled (nerfhip_set_profiling(h,1)). */
typedef struct nerfhip_prof {
  float mlp_fwd_ms;      /* fused posenc+MLP forward kernels (all launches of the call) */
  float mlp_bwd_ms;      /* MLP backward-data kernels */
  float mlp_wgrad_ms;    /* weight-gradient GEMM kernel */
  float composite_ms;    /* compositing fwd(+bwd) */
  float sampling_ms;     /* stratified + inverse-CDF */
  float other_ms;        /* pack, loss, adam, dir branch */
  int32_t mlp_fwd_launches;
  int32_t reserved;
  double mlp_fwd_flop;   /* algorithmic FLOP of those launches (SURVEY 8d figures) */
  double mlp_bwd_flop;
  double mlp_wgrad_flop;
  float comm_ms;         /* gradient all-reduce (nerfhip_dp_allreduce_grads) */
  float reserved2;
} nerfhip_prof;
int nerfhip_set_profiling(nerfhip_t* h, int32_t on);
int nerfhip_get_profile(nerfhip_t* h, nerfhip_prof* out);

const char* nerfhip_version(void);

#ifdef __cplusplus
}
#endif
#endif
