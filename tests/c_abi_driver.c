/* c_abi_driver.c -- a plain-C (C11) client of include/nerfhip.h, the closest stand-in available here for the JNI stub
 * a maintainer would put under the reference's Scala entry points (nerfBlock.scala:146-157 forwardInternal,
 * runNerf.scala:100-106 train step, :181-204 renderToImage).  No Python, no torch, no C++:
 * it proves the header is valid C, that the structs have the layout the other bindings assume, and that the call
 * sequence create -> set_params -> render -> render_u8 -> render_pose(_rows)_u8 -> train_step -> get_params -> destroy
 * works from C.
 *
 *   c_abi_driver layout            prints sizeof / offsetof of every struct of the ABI (no GPU needed)
 *   c_abi_driver run <dir>         reads <dir>/config.bin (raw struct nerfhip_config), params.bin, rays_o.bin,
 *                                  rays_d.bin, bounds.bin, label.bin (raw little-endian fp32, [n][.]); writes
 *                                  rgb.bin, u8.bin, depth.bin, pose.bin, loss.bin, params_after.bin, grads.bin
 * Built by __graft_entry__.build() with gcc -std=c11 -Wall -Wextra -pedantic. */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nerfhip.h"

#define OFF(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))

static int layout(void) {
  printf("sizeof.nerfhip_config %zu\n", sizeof(nerfhip_config));
  OFF(nerfhip_config, use_dir); OFF(nerfhip_config, use_sh); OFF(nerfhip_config, use_hierarchical);
  OFF(nerfhip_config, pos_l); OFF(nerfhip_config, dir_l); OFF(nerfhip_config, D); OFF(nerfhip_config, W);
  OFF(nerfhip_config, n_skips); OFF(nerfhip_config, skips); OFF(nerfhip_config, n_samples);
  OFF(nerfhip_config, n_importance); OFF(nerfhip_config, raw_noise_std); OFF(nerfhip_config, white_bkgd);
  OFF(nerfhip_config, lin_disp); OFF(nerfhip_config, perturb); OFF(nerfhip_config, lrate);
  OFF(nerfhip_config, lrate_decay); OFF(nerfhip_config, device); OFF(nerfhip_config, dtype);
  OFF(nerfhip_config, seed); OFF(nerfhip_config, max_rays); OFF(nerfhip_config, reserved);
  printf("sizeof.nerfhip_aux %zu\n", sizeof(nerfhip_aux));
  OFF(nerfhip_aux, t_rand); OFF(nerfhip_aux, u_rand); OFF(nerfhip_aux, noise_c); OFF(nerfhip_aux, noise_f);
  OFF(nerfhip_aux, z_coarse); OFF(nerfhip_aux, w_coarse); OFF(nerfhip_aux, z_fine); OFF(nerfhip_aux, w_fine);
  OFF(nerfhip_aux, sigma_c); OFF(nerfhip_aux, sigma_f); OFF(nerfhip_aux, craw_c); OFF(nerfhip_aux, craw_f);
  OFF(nerfhip_aux, depth); OFF(nerfhip_aux, acc);
  printf("sizeof.nerfhip_prof %zu\n", sizeof(nerfhip_prof));
  OFF(nerfhip_prof, mlp_fwd_ms); OFF(nerfhip_prof, mlp_bwd_ms); OFF(nerfhip_prof, mlp_wgrad_ms);
  OFF(nerfhip_prof, composite_ms); OFF(nerfhip_prof, sampling_ms); OFF(nerfhip_prof, other_ms);
  OFF(nerfhip_prof, mlp_fwd_launches); OFF(nerfhip_prof, reserved); OFF(nerfhip_prof, mlp_fwd_flop);
  OFF(nerfhip_prof, mlp_bwd_flop); OFF(nerfhip_prof, mlp_wgrad_flop); OFF(nerfhip_prof, comm_ms);
  OFF(nerfhip_prof, reserved2);
  printf("const.NERFHIP_MAX_SKIPS %d\n", NERFHIP_MAX_SKIPS);
  printf("const.NERFHIP_DP_ID_BYTES %d\n", NERFHIP_DP_ID_BYTES);
  printf("const.NERFHIP_E_COMM %d\n", NERFHIP_E_COMM);
  printf("version %s\n", nerfhip_version());
  return 0;
}

static void* slurp(const char* dir, const char* name, size_t* bytes) {
  char path[4096];
  snprintf(path, sizeof path, "%s/%s", dir, name);
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long len = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(len > 0 ? (size_t)len : 1);
  if (!p || fread(p, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  fclose(f);
  *bytes = (size_t)len;
  return p;
}

static void dump(const char* dir, const char* name, const void* p, size_t bytes) {
  char path[4096];
  snprintf(path, sizeof path, "%s/%s", dir, name);
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
  fclose(f);
}

#define CHK(h, call)                                                                      \
  do { int rc_ = (call); if (rc_ != NERFHIP_OK) {                                          \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nerfhip_last_error(h)); return 1; } } while (0)

static int run(const char* dir) {
  size_t nb = 0;
  nerfhip_config* cfg = (nerfhip_config*)slurp(dir, "config.bin", &nb);
  if (nb != sizeof(nerfhip_config)) { fprintf(stderr, "config.bin has %zu bytes, struct has %zu\n", nb, sizeof *cfg); return 2; }
  float* params = (float*)slurp(dir, "params.bin", &nb);
  const int64_t np = (int64_t)(nb / sizeof(float));
  float* ro = (float*)slurp(dir, "rays_o.bin", &nb);
  const int64_t n = (int64_t)(nb / (3 * sizeof(float)));
  float* rd = (float*)slurp(dir, "rays_d.bin", &nb);
  float* bd = (float*)slurp(dir, "bounds.bin", &nb);
  float* label = (float*)slurp(dir, "label.bin", &nb);

  nerfhip_t* h = NULL;
  CHK(NULL, nerfhip_create(cfg, &h));
  if (nerfhip_param_count(h) != np) { fprintf(stderr, "params.bin holds %lld floats, the model %lld\n", (long long)np, (long long)nerfhip_param_count(h)); return 2; }
  CHK(h, nerfhip_set_params(h, params, np));

  /* trainer.evaluate(NDList(raysO, raysD, bounds, viewDir = raysD)) (runNerf.scala:192-194) */
  float* rgb = (float*)malloc((size_t)n * 3 * sizeof(float));
  float* depth = (float*)malloc((size_t)n * sizeof(float));
  uint8_t* u8 = (uint8_t*)malloc((size_t)n * 3);
  nerfhip_aux aux;
  memset(&aux, 0, sizeof aux);
  aux.depth = depth;
  CHK(h, nerfhip_render(h, ro, rd, bd, rd, n, rgb, &aux));
  CHK(h, nerfhip_render_u8(h, ro, rd, bd, rd, n, u8));
  dump(dir, "rgb.bin", rgb, (size_t)n * 3 * sizeof(float));
  dump(dir, "depth.bin", depth, (size_t)n * sizeof(float));
  dump(dir, "u8.bin", u8, (size_t)n * 3);

  /* renderToImage from a pose (runNerf.scala:181-204), whole and as three row blocks of unequal height -- the shape of a
   * frame sharded over GPUs (BASELINE configs[2]); stacked, the blocks must be the whole frame byte for byte */
  {
    enum { PH = 26, PW = 10 };
    const float c2w[12] = {1.f, 0.f, 0.f, 0.2f, 0.f, 0.6f, -0.8f, -3.2f, 0.f, 0.8f, 0.6f, 2.4f};
    const float focal = 22.0f;
    uint8_t whole[PH * PW * 3], rows[PH * PW * 3];
    const int cut[4] = {0, 9, 18, PH};
    CHK(h, nerfhip_render_pose_u8(h, PH, PW, focal, c2w, 2.0f, 6.0f, 0, whole));
    for (int k = 0; k < 3; ++k)
      CHK(h, nerfhip_render_pose_rows_u8(h, PH, PW, focal, c2w, 2.0f, 6.0f, 0, cut[k], cut[k + 1], rows + (size_t)cut[k] * PW * 3));
    if (memcmp(whole, rows, sizeof whole) != 0) { fprintf(stderr, "row blocks differ from the whole frame\n"); return 1; }
    CHK(h, nerfhip_render_pose_rows_u8(h, PH, PW, focal, c2w, 2.0f, 6.0f, 0, 7, 7, NULL));   /* a rank without rows */
    if (nerfhip_render_pose_rows_u8(h, PH, PW, focal, c2w, 2.0f, 6.0f, 0, 3, PH + 1, rows) != NERFHIP_E_INVALID) {
      fprintf(stderr, "rows past the image were not refused\n"); return 1; }
    dump(dir, "pose.bin", whole, sizeof whole);
  }

  /* one training iteration (runNerf.scala:94-106), gradients read back before the optimizer consumes them */
  float loss[2] = {0.f, 0.f};
  float* grads = (float*)malloc((size_t)np * sizeof(float));
  CHK(h, nerfhip_train_backward(h, ro, rd, bd, rd, label, n, loss, NULL));
  CHK(h, nerfhip_get_grads(h, grads, np));
  CHK(h, nerfhip_optimizer_step(h, 1.0f));
  CHK(h, nerfhip_synchronize(h));
  CHK(h, nerfhip_get_params(h, params, np));
  dump(dir, "loss.bin", loss, sizeof loss);
  dump(dir, "grads.bin", grads, (size_t)np * sizeof(float));
  dump(dir, "params_after.bin", params, (size_t)np * sizeof(float));

  /* error path from C: an oversized batch is refused with a code and a message, not a crash */
  if (nerfhip_train_backward(h, ro, rd, bd, rd, label, (int64_t)cfg->max_rays + 1, loss, NULL) != NERFHIP_E_INVALID ||
      strlen(nerfhip_last_error(h)) == 0) { fprintf(stderr, "oversized batch was not refused\n"); return 1; }
  nerfhip_destroy(h);
  free(cfg); free(params); free(ro); free(rd); free(bd); free(label); free(rgb); free(depth); free(u8); free(grads);
  printf("c_abi_driver ok: %lld rays, loss %.9g %.9g\n", (long long)n, (double)loss[0], (double)loss[1]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && strcmp(argv[1], "layout") == 0) return layout();
  if (argc >= 3 && strcmp(argv[1], "run") == 0) return run(argv[2]);
  fprintf(stderr, "usage: %s layout | run <dir>\n", argv[0]);
  return 2;
}
