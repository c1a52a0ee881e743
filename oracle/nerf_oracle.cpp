/*
 * nerf_oracle.cpp -- CPU restatement of the reference NeRF hot path.  TEST INFRASTRUCTURE ONLY:
 * see nerf_oracle.h for the scope, the citation list and the "parity unpinned" statement.
 *
 * Conventions fixed here (build definitions where the reference defers to libtorch):
 *   - linspace is libtorch's symmetric form (i < n/2: i*step, else fma(-step, n-1-i, 1)), fp32 step;
 *   - cumsum over fp32 accumulates in double and rounds every element to fp32 (libtorch CPU);
 *   - bin index = #{cdf <= u} (stable double-argsort == searchsorted(right=True));
 *   - Linear is y = x W^T + b with W stored [out][in] (DJL Linear).
 * Plain libm sinf/cosf/expf/sqrtf; fp32 throughout except where stated.
 */
#include "nerf_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ----------------------------------------------------------------------------------------------
// parameter layout: DJL child order (nerfBlock.scala:16-19, nerfCoreBlock.scala:19-48,52-85)
// ----------------------------------------------------------------------------------------------
struct NetLayout {
  int D, W, E, Ed, H2;          // H2 = W/2
  int mode;                     // 0: Linear(3) | 1: view-dir head | 2: SH head
  int fan_in[ORC_MAX_D];
  int64_t w_off[ORC_MAX_D], b_off[ORC_MAX_D];
  int64_t ws_off, bs_off;       // density head
  int64_t wf_off, bf_off;       // mode 1: Linear(W/2) on features | mode 2: Linear(27) | mode 0: Linear(3)
  int64_t wd_off, bd_off;       // mode 1: Linear(W/2) on enc(dir)
  int64_t w3_off, b3_off;       // mode 1: Linear(3)
  int64_t count;
};

bool is_skip(const orc_config* c, int i) {
  for (int k = 0; k < c->n_skips; ++k)
    if (c->skips[k] == i) return true;
  return false;
}

NetLayout make_layout(const orc_config* c) {
  NetLayout L{};
  L.D = c->D; L.W = c->W; L.E = 3 * (1 + 2 * c->pos_l); L.Ed = 3 * (1 + 2 * c->dir_l); L.H2 = c->W / 2;
  L.mode = c->use_dir ? (c->use_sh ? 2 : 1) : 0;
  int64_t off = 0;
  for (int l = 0; l < L.D; ++l) {
    L.fan_in[l] = (l == 0) ? L.E : (L.W + (is_skip(c, l - 1) ? L.E : 0));
    L.w_off[l] = off; off += (int64_t)L.W * L.fan_in[l];
    L.b_off[l] = off; off += L.W;
  }
  // NOTE: a skip after the LAST trunk layer widens the heads' input; the reference would do the same
  // (nerfCoreBlock.scala:21-31).  Supported via feat_w below.
  int feat_w = L.W + (is_skip(c, L.D - 1) ? L.E : 0);
  L.ws_off = off; off += feat_w; L.bs_off = off; off += 1;
  if (L.mode == 1) {
    L.wf_off = off; off += (int64_t)L.H2 * feat_w; L.bf_off = off; off += L.H2;
    L.wd_off = off; off += (int64_t)L.H2 * L.Ed;   L.bd_off = off; off += L.H2;
    L.w3_off = off; off += 3 * L.H2;               L.b3_off = off; off += 3;
  } else if (L.mode == 2) {
    L.wf_off = off; off += (int64_t)27 * feat_w; L.bf_off = off; off += 27;
  } else {
    L.wf_off = off; off += (int64_t)3 * feat_w; L.bf_off = off; off += 3;
  }
  L.count = off;
  return L;
}

// ----------------------------------------------------------------------------------------------
// C[M][N] (+)= A[M][K] * B[K][N], row-major, fp32; register-blocked so gcc vectorises over N.
// ----------------------------------------------------------------------------------------------
constexpr int MR = 6, NR = 16;

void gemm_nn(int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
             int64_t ldc, bool accumulate) {
  const int64_t mblocks = (M + MR - 1) / MR;
#pragma omp parallel for schedule(static)
  for (int64_t mb = 0; mb < mblocks; ++mb) {
    const int64_t m0 = mb * MR;
    const int mr = (int)std::min<int64_t>(MR, M - m0);
    for (int n0 = 0; n0 < N; n0 += NR) {
      const int nr = std::min(NR, N - n0);
      float acc[MR][NR];
      for (int r = 0; r < MR; ++r)
        for (int j = 0; j < NR; ++j)
          acc[r][j] = (accumulate && r < mr && j < nr) ? C[(m0 + r) * ldc + n0 + j] : 0.f;
      if (mr == MR && nr == NR) {
        const float* a0 = A + m0 * lda;
        for (int k = 0; k < K; ++k) {
          const float* b = B + (int64_t)k * ldb + n0;
          for (int r = 0; r < MR; ++r) {
            const float a = a0[r * lda + k];
            for (int j = 0; j < NR; ++j) acc[r][j] += a * b[j];
          }
        }
      } else {
        for (int k = 0; k < K; ++k) {
          const float* b = B + (int64_t)k * ldb + n0;
          for (int r = 0; r < mr; ++r) {
            const float a = A[(m0 + r) * lda + k];
            for (int j = 0; j < nr; ++j) acc[r][j] += a * b[j];
          }
        }
      }
      for (int r = 0; r < mr; ++r)
        for (int j = 0; j < nr; ++j) C[(m0 + r) * ldc + n0 + j] = acc[r][j];
    }
  }
}

void transpose(const float* in, int64_t rows, int64_t cols, float* out) {  // out[cols][rows]
#pragma omp parallel for schedule(static)
  for (int64_t c0 = 0; c0 < cols; c0 += 32)
    for (int64_t r0 = 0; r0 < rows; r0 += 32)
      for (int64_t c = c0; c < std::min(cols, c0 + 32); ++c)
        for (int64_t r = r0; r < std::min(rows, r0 + 32); ++r) out[c * rows + r] = in[r * cols + c];
}

// Linear forward: Y[M][N] = X[M][K] W^T + b, W is [N][K].
void linear_fwd(int64_t M, int N, int K, const float* X, int64_t ldx, const float* Wm, const float* b,
                float* Y, int64_t ldy) {
  std::vector<float> Wt((size_t)K * N);
  transpose(Wm, N, K, Wt.data());
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) Y[m * ldy + n] = b ? b[n] : 0.f;
  gemm_nn(M, N, K, X, ldx, Wt.data(), N, Y, ldy, true);
}
// dX[M][K] (+)= dY[M][N] W ; W is [N][K] with leading dim ldw (lets callers pick a column slice).
void linear_bwd_x(int64_t M, int N, int K, const float* dY, int64_t ldy, const float* Wm, int64_t ldw,
                  float* dX, int64_t ldx, bool accumulate) {
  gemm_nn(M, K, N, dY, ldy, Wm, ldw, dX, ldx, accumulate);
}
// dW[N][K] += dY^T X ; db[N] += colsum(dY)
void linear_bwd_w(int64_t M, int N, int K, const float* dY, int64_t ldy, const float* X, int64_t ldx,
                  float* dW, int64_t ldw, float* db) {
  std::vector<float> dYt((size_t)N * M);
  if (ldy == N) transpose(dY, M, N, dYt.data());
  else {
    for (int64_t m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) dYt[(size_t)n * M + m] = dY[m * ldy + n];
  }
  gemm_nn(N, K, (int)M, dYt.data(), M, X, ldx, dW, ldw, true);
  if (db) {
    for (int n = 0; n < N; ++n) {
      double s = 0;
      const float* row = dYt.data() + (size_t)n * M;
      for (int64_t m = 0; m < M; ++m) s += row[m];
      db[n] += (float)s;
    }
  }
}

inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// nerfCoreBlock.scala:127-136
void position_code_row(const float* x, int L, float* out) {
  out[0] = x[0]; out[1] = x[1]; out[2] = x[2];
  for (int i = 0; i < L; ++i) {
    const float f = (float)(1 << i);
    for (int k = 0; k < 3; ++k) {
      const float a = x[k] * f;
      out[3 + 6 * i + k] = sinf(a);
      out[3 + 6 * i + 3 + k] = cosf(a);
    }
  }
}

// nerfCoreBlock.scala:138-174
void sh2_row(const float* v, float* Y) {
  const float x = v[0], y = v[1], z = v[2];
  const float cosPhi = z;
  const float sinPhi = sqrtf(-(z * z - 1.0f));
  const bool small = sinPhi < 1e-5f;
  const float cosTheta = small ? 1.0f : x / sinPhi;
  const float sinTheta = small ? 0.0f : y / sinPhi;
  const float stcp = sinTheta * cosPhi;
  const float stsp = sinTheta * sinPhi;
  Y[0] = 1.0f;
  Y[1] = cosTheta;
  Y[2] = stcp;
  Y[3] = stsp;
  Y[4] = cosTheta * cosTheta - (float)(1.0 / 3);
  Y[5] = stcp * cosTheta;
  Y[6] = stsp * cosTheta;
  Y[7] = (stcp * cosPhi * 2.0f - sinTheta) * sinTheta;
  Y[8] = stsp * sinTheta * cosPhi;
}

// Activations kept for the backward pass of one network.
struct NetCache {
  int64_t M = 0;
  int feat_w = 0;
  std::vector<float> enc;                 // [M][E]
  std::vector<std::vector<float>> in;     // in[l]: input of trunk layer l, [M][fan_in[l]]
  std::vector<float> feat;                // [M][feat_w]  (output of the trunk incl. a trailing skip)
  std::vector<float> hc;                  // mode 1: [M][H2] post-relu hidden; mode 2: [M][27] q
  std::vector<float> dirfeat;             // mode 1: enc(dir) [B][Ed]; mode 2: SH [B][9]
};

// nerfCoreBlock.forwardInternal (nerfCoreBlock.scala:107-112) on M = B*S samples, ray-major.
void net_forward(const orc_config* c, const NetLayout& L, const float* P, bool want_rgb, int64_t B, int S,
                 const float* rays_o, const float* rays_d, const float* viewdir, const float* z,
                 float* sigma, float* craw, NetCache* cache) {
  const int64_t M = B * S;
  const int E = L.E, W = L.W;
  NetCache local;
  NetCache& C = cache ? *cache : local;
  C.M = M;
  C.enc.resize((size_t)M * E);
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    const int64_t b = m / S;
    float pos[3];
    for (int k = 0; k < 3; ++k) pos[k] = rays_o[b * 3 + k] + rays_d[b * 3 + k] * z[m];  // nerfBlock.scala:65
    position_code_row(pos, c->pos_l, &C.enc[(size_t)m * E]);
  }
  C.in.assign(L.D, {});
  std::vector<float> cur = C.enc;  // input of layer 0
  int cur_w = E;
  for (int l = 0; l < L.D; ++l) {
    std::vector<float> out((size_t)M * W);
    linear_fwd(M, W, cur_w, cur.data(), cur_w, P + L.w_off[l], P + L.b_off[l], out.data(), W);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * W; ++i) out[i] = out[i] > 0.f ? out[i] : 0.f;
    C.in[l] = std::move(cur);
    if (is_skip(c, l)) {  // concat(h, enc): h first (nerfCoreBlock.scala:24)
      cur.assign((size_t)M * (W + E), 0.f);
#pragma omp parallel for schedule(static)
      for (int64_t m = 0; m < M; ++m) {
        memcpy(&cur[(size_t)m * (W + E)], &out[(size_t)m * W], sizeof(float) * W);
        memcpy(&cur[(size_t)m * (W + E) + W], &C.enc[(size_t)m * E], sizeof(float) * E);
      }
      cur_w = W + E;
    } else {
      cur = std::move(out);
      cur_w = W;
    }
  }
  C.feat_w = cur_w;
  // density head on samples 0..S-2 of every ray (nerfCoreBlock.scala:39-40); column S-1 := 0
  {
    const float* ws = P + L.ws_off;
    const float bs = P[L.bs_off];
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
      if ((m % S) == S - 1) { sigma[m] = 0.f; continue; }
      float acc = 0.f;
      const float* f = &cur[(size_t)m * cur_w];
      for (int k = 0; k < cur_w; ++k) acc += f[k] * ws[k];
      sigma[m] = acc + bs;
    }
  }
  if (want_rgb && craw) {
    if (L.mode == 1) {
      const int H2 = L.H2, Ed = L.Ed;
      C.dirfeat.resize((size_t)B * Ed);
      for (int64_t b = 0; b < B; ++b) position_code_row(viewdir + b * 3, c->dir_l, &C.dirfeat[(size_t)b * Ed]);
      std::vector<float> dv((size_t)B * H2);
      linear_fwd(B, H2, Ed, C.dirfeat.data(), Ed, P + L.wd_off, P + L.bd_off, dv.data(), H2);
      C.hc.resize((size_t)M * H2);
      linear_fwd(M, H2, cur_w, cur.data(), cur_w, P + L.wf_off, P + L.bf_off, C.hc.data(), H2);
#pragma omp parallel for schedule(static)
      for (int64_t m = 0; m < M; ++m) {
        const int64_t b = m / S;
        float* h = &C.hc[(size_t)m * H2];
        for (int j = 0; j < H2; ++j) {  // add then relu (nerfCoreBlock.scala:67)
          const float v = h[j] + dv[(size_t)b * H2 + j];
          h[j] = v > 0.f ? v : 0.f;
        }
      }
      linear_fwd(M, 3, H2, C.hc.data(), H2, P + L.w3_off, P + L.b3_off, craw, 3);
    } else if (L.mode == 2) {
      C.dirfeat.resize((size_t)B * 9);
      for (int64_t b = 0; b < B; ++b) sh2_row(viewdir + b * 3, &C.dirfeat[(size_t)b * 9]);
      C.hc.resize((size_t)M * 27);
      linear_fwd(M, 27, cur_w, cur.data(), cur_w, P + L.wf_off, P + L.bf_off, C.hc.data(), 27);
#pragma omp parallel for schedule(static)
      for (int64_t m = 0; m < M; ++m) {
        const int64_t b = m / S;
        const float* q = &C.hc[(size_t)m * 27];
        const float* Y = &C.dirfeat[(size_t)b * 9];
        for (int k = 0; k < 3; ++k) {  // reshape(...,9,3) * SH[...,None] summed over 9 (nerfCoreBlock.scala:55)
          float s = 0.f;
          for (int j = 0; j < 9; ++j) s += q[j * 3 + k] * Y[j];
          craw[m * 3 + k] = s;
        }
      }
    } else {
      linear_fwd(M, 3, cur_w, cur.data(), cur_w, P + L.wf_off, P + L.bf_off, craw, 3);
    }
  }
  C.feat = std::move(cur);
}

// Reverse-mode of net_forward.  d_sigma [M] (column S-1 ignored), d_craw [M][3] or NULL.  G += grads.
void net_backward(const orc_config* /*c*/, const NetLayout& L, const float* P, int64_t B, int S,
                  const NetCache& C, const float* d_sigma, const float* d_craw, float* G) {
  const int64_t M = B * S;
  const int W = L.W, E = L.E, fw = C.feat_w;
  std::vector<float> dfeat((size_t)M * fw, 0.f);
  if (d_craw) {
    if (L.mode == 1) {
      const int H2 = L.H2, Ed = L.Ed;
      linear_bwd_w(M, 3, H2, d_craw, 3, C.hc.data(), H2, G + L.w3_off, H2, G + L.b3_off);
      std::vector<float> dh((size_t)M * H2);
      linear_bwd_x(M, 3, H2, d_craw, 3, P + L.w3_off, H2, dh.data(), H2, false);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < M * H2; ++i) dh[i] = C.hc[i] > 0.f ? dh[i] : 0.f;
      linear_bwd_w(M, H2, fw, dh.data(), H2, C.feat.data(), fw, G + L.wf_off, fw, G + L.bf_off);
      linear_bwd_x(M, H2, fw, dh.data(), H2, P + L.wf_off, fw, dfeat.data(), fw, true);
      std::vector<float> ddv((size_t)B * H2, 0.f);  // the per-ray dir branch is broadcast over samples
#pragma omp parallel for schedule(static)
      for (int64_t b = 0; b < B; ++b)
        for (int s = 0; s < S; ++s)
          for (int j = 0; j < H2; ++j) ddv[(size_t)b * H2 + j] += dh[(size_t)(b * S + s) * H2 + j];
      linear_bwd_w(B, H2, Ed, ddv.data(), H2, C.dirfeat.data(), Ed, G + L.wd_off, Ed, G + L.bd_off);
    } else if (L.mode == 2) {
      std::vector<float> dq((size_t)M * 27);
#pragma omp parallel for schedule(static)
      for (int64_t m = 0; m < M; ++m) {
        const float* Y = &C.dirfeat[(size_t)(m / S) * 9];
        for (int j = 0; j < 9; ++j)
          for (int k = 0; k < 3; ++k) dq[(size_t)m * 27 + j * 3 + k] = d_craw[m * 3 + k] * Y[j];
      }
      linear_bwd_w(M, 27, fw, dq.data(), 27, C.feat.data(), fw, G + L.wf_off, fw, G + L.bf_off);
      linear_bwd_x(M, 27, fw, dq.data(), 27, P + L.wf_off, fw, dfeat.data(), fw, true);
    } else {
      linear_bwd_w(M, 3, fw, d_craw, 3, C.feat.data(), fw, G + L.wf_off, fw, G + L.bf_off);
      linear_bwd_x(M, 3, fw, d_craw, 3, P + L.wf_off, fw, dfeat.data(), fw, true);
    }
  }
  {  // density head
    std::vector<float> ds(M);
    for (int64_t m = 0; m < M; ++m) ds[m] = ((m % S) == S - 1) ? 0.f : d_sigma[m];
    linear_bwd_w(M, 1, fw, ds.data(), 1, C.feat.data(), fw, G + L.ws_off, fw, G + L.bs_off);
    linear_bwd_x(M, 1, fw, ds.data(), 1, P + L.ws_off, fw, dfeat.data(), fw, true);
  }
  // trunk, last layer first.  dcur is the gradient w.r.t. the (possibly concatenated) input of layer l+1.
  std::vector<float> dcur = std::move(dfeat);
  int dcur_w = fw;
  for (int l = L.D - 1; l >= 0; --l) {
    // output of layer l (post relu) = first W columns of the next layer's input
    const float* outp = (l == L.D - 1) ? C.feat.data() : C.in[l + 1].data();
    const int outp_w = (l == L.D - 1) ? fw : L.fan_in[l + 1];
    std::vector<float> dy((size_t)M * W);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
      for (int j = 0; j < W; ++j)
        dy[(size_t)m * W + j] = outp[(size_t)m * outp_w + j] > 0.f ? dcur[(size_t)m * dcur_w + j] : 0.f;
    const int K = L.fan_in[l];
    linear_bwd_w(M, W, K, dy.data(), W, C.in[l].data(), K, G + L.w_off[l], K, G + L.b_off[l]);
    if (l > 0) {
      dcur.assign((size_t)M * K, 0.f);
      dcur_w = K;
      // the encoding part of a skip input needs no gradient (positions are not trainable)
      linear_bwd_x(M, W, K, dy.data(), W, P + L.w_off[l], K, dcur.data(), K, false);
    }
  }
  (void)E;
}

// nerfBlock.getWeight + weighted sum + addBkgd for ONE ray (nerfBlock.scala:24-45,115).
void composite_ray(const orc_config* c, int S, const float* sigma, const float* noise, const float* craw,
                   const float* z, float* w, float* rgb, float* depth, float* acc) {
  double run = 0.0;  // cumSum(0): double accumulator, every element rounded to fp32
  float Tprev = 1.0f;
  for (int i = 0; i < S - 1; ++i) {
    const float dist = z[i + 1] - z[i];
    float s = sigma[i];
    if (noise) s = s + noise[i] * c->raw_noise_std;  // addNoise (nerfBlock.scala:21)
    const float r = s > 0.f ? s : 0.f;
    const float a = (-r) * dist;
    run += (double)a;
    const float T = expf((float)run);
    const float alpha = -(expf(a) - 1.0f);
    w[i] = (i == 0) ? alpha : Tprev * alpha;
    Tprev = T;
  }
  w[S - 1] = Tprev;
  float sw = 0.f, sz = 0.f, col[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < S; ++i) {
    sw += w[i];
    sz += w[i] * z[i];
    if (craw)
      for (int k = 0; k < 3; ++k) col[k] += w[i] * sigmoidf(craw[i * 3 + k]);
  }
  if (rgb && craw)
    for (int k = 0; k < 3; ++k) rgb[k] = c->white_bkgd ? col[k] + (-(sw - 1.0f)) : col[k];
  if (depth) *depth = sz;
  if (acc) *acc = sw;
}

void composite_bwd_ray(const orc_config* c, int S, const float* sigma, const float* noise, const float* craw,
                       const float* z, const float* w, const float* d_rgb, float* d_sigma, float* d_craw) {
  // recompute a_i, e_i = exp(a_i), T_i
  std::vector<float> a(S), T(S), e(S), dw(S);
  double run = 0.0;
  for (int i = 0; i < S - 1; ++i) {
    const float dist = z[i + 1] - z[i];
    float s = sigma[i];
    if (noise) s = s + noise[i] * c->raw_noise_std;
    const float r = s > 0.f ? s : 0.f;
    a[i] = (-r) * dist;
    run += (double)a[i];
    T[i] = expf((float)run);
    e[i] = expf(a[i]);
  }
  const float dsum = d_rgb[0] + d_rgb[1] + d_rgb[2];
  for (int i = 0; i < S; ++i) {
    float g = 0.f;
    for (int k = 0; k < 3; ++k) {
      const float sg = sigmoidf(craw[i * 3 + k]);
      g += d_rgb[k] * sg;
      d_craw[i * 3 + k] = w[i] * d_rgb[k] * sg * (1.0f - sg);
    }
    dw[i] = c->white_bkgd ? g - dsum : g;
  }
  // w_0 = alpha_0 ; w_i = T_{i-1} alpha_i ; w_{S-1} = T_{S-2}
  double suffix = 0.0;  // sum_{i>=j} dA_i with dA_i = dT_i * T_i
  for (int j = S - 2; j >= 0; --j) {
    const float dT = (j == S - 2) ? dw[S - 1] : dw[j + 1] * (1.0f - e[j + 1]);
    suffix += (double)dT * (double)T[j];
    const float dalpha = dw[j] * (j == 0 ? 1.0f : T[j - 1]);
    const float da = (float)suffix + dalpha * (-e[j]);
    const float dist = z[j + 1] - z[j];
    float s = sigma[j];
    if (noise) s = s + noise[j] * c->raw_noise_std;
    d_sigma[j] = s > 0.f ? (-dist) * da : 0.f;
  }
  d_sigma[S - 1] = 0.f;
}

// nerfBlock.samplePdf for ONE ray (nerfBlock.scala:74-106).
void sample_pdf_ray(int N, int NI, const float* w, const float* z, const float* u_sorted, float* cdf,
                    int32_t* ind, float* zs, float* zf) {
  const int nb = N - 1;  // cdf entries == number of bins (midpoints)
  std::vector<float> wc(N - 2), bins(nb), cdf_local(nb);
  float* cd = cdf ? cdf : cdf_local.data();
  double run = 0.0;
  for (int k = 0; k < N - 2; ++k) {
    const float t = w[k + 1] + 1e-5f;  // weight[1:-1] + 1e-5 (fp32 add)
    run += (double)t;
    wc[k] = (float)run;
  }
  const float sum = wc[N - 3];
  cd[0] = 0.f;
  for (int k = 0; k < N - 2; ++k) cd[k + 1] = wc[k] / sum;
  for (int k = 0; k < nb; ++k) bins[k] = (z[k + 1] + z[k]) * 0.5f;
  std::vector<float> tmp(N + NI);
  for (int m = 0; m < NI; ++m) {
    const float u = u_sorted[m];
    int id = 0;  // #{cdf <= u}
    while (id < nb && cd[id] <= u) ++id;
    const int below = std::max(id - 1, 0);
    const int above = std::min(id, nb - 1);
    float den = cd[above] - cd[below];
    if (den < 1e-5f) den = 1.0f;
    const float t = (u - cd[below]) / den;
    const float s = bins[below] + t * (bins[above] - bins[below]);
    if (ind) ind[m] = id;
    if (zs) zs[m] = s;
    tmp[m] = s;
  }
  for (int i = 0; i < N; ++i) tmp[NI + i] = z[i];
  std::sort(tmp.begin(), tmp.end());
  memcpy(zf, tmp.data(), sizeof(float) * (N + NI));
}

struct SplitMix {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
  double normal() {
    const double u1 = uniform(), u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

}  // namespace

extern "C" {

int32_t orc_pos_width(const orc_config* c) { return 3 * (1 + 2 * c->pos_l); }
int64_t orc_net_param_count(const orc_config* c) { return make_layout(c).count; }
int64_t orc_param_count(const orc_config* c) {
  return make_layout(c).count * (c->use_hierarchical ? 2 : 1);
}
int32_t orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_linspace(int32_t n, float* out) {
  if (n == 1) { out[0] = 0.f; return; }
  const float step = (1.0f - 0.0f) / (float)(n - 1);
  const int half = n / 2;
  // torch CPU evaluates `end - step*(n-1-i)` with ONE rounding (observed with torch 2.10: identical to a
  // double-precision evaluation of the fp32 step), i.e. an fma; the first half is a plain fp32 product.
  for (int i = 0; i < n; ++i) out[i] = (i < half) ? step * (float)i : std::fmaf(-step, (float)(n - i - 1), 1.0f);
}

void orc_position_code(const float* x, int64_t n, int32_t L, float* out) {
  const int E = 3 * (1 + 2 * L);
  for (int64_t i = 0; i < n; ++i) position_code_row(x + i * 3, L, out + i * E);
}

void orc_sh2(const float* dir, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) sh2_row(dir + i * 3, out + i * 9);
}

// nerfBlock.getSample + givePerturb (nerfBlock.scala:47-57,63-64)
void orc_stratified(const orc_config* c, int64_t B, const float* bounds, const float* t_rand, float* z) {
  const int N = c->n_samples;
  std::vector<float> t(N);
  orc_linspace(N, t.data());
  for (int64_t b = 0; b < B; ++b) {
    const float near = bounds[b * 2], far = bounds[b * 2 + 1];
    float* zr = z + b * N;
    for (int i = 0; i < N; ++i) {
      const float omt = -(t[i] - 1.0f);
      if (c->lin_disp) zr[i] = 1.0f / ((1.0f / near) * omt + (1.0f / far) * t[i]);
      else zr[i] = near * omt + far * t[i];
    }
    if (c->perturb && t_rand) {
      std::vector<float> lower(N), upper(N);
      for (int i = 0; i < N - 1; ++i) {
        const float mid = (zr[i + 1] + zr[i]) * 0.5f;
        upper[i] = mid;
        lower[i + 1] = mid;
      }
      upper[N - 1] = zr[N - 1];
      lower[0] = zr[0];
      for (int i = 0; i < N; ++i) zr[i] = lower[i] + (upper[i] - lower[i]) * t_rand[b * N + i];
    }
  }
}

void orc_composite(const orc_config* c, int64_t B, int32_t S, const float* sigma, const float* noise,
                   const float* craw, const float* z, float* w, float* rgb, float* depth, float* acc) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b)
    composite_ray(c, S, sigma + b * S, noise ? noise + b * S : nullptr, craw ? craw + b * S * 3 : nullptr,
                  z + b * S, w + b * S, rgb ? rgb + b * 3 : nullptr, depth ? depth + b : nullptr,
                  acc ? acc + b : nullptr);
}

void orc_composite_bwd(const orc_config* c, int64_t B, int32_t S, const float* sigma, const float* noise,
                       const float* craw, const float* z, const float* w, const float* d_rgb,
                       float* d_sigma, float* d_craw) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b)
    composite_bwd_ray(c, S, sigma + b * S, noise ? noise + b * S : nullptr, craw + b * S * 3, z + b * S,
                      w + b * S, d_rgb + b * 3, d_sigma + b * S, d_craw + b * S * 3);
}

void orc_sample_pdf(const orc_config* c, int64_t B, const float* w, const float* z, const float* u_rand,
                    float* cdf, int32_t* ind, float* z_samples, float* z_fine) {
  const int N = c->n_samples, NI = c->n_importance;
  std::vector<float> lin(NI);
  orc_linspace(NI, lin.data());
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    std::vector<float> u(lin);
    if (c->perturb && u_rand) {  // randomUniform(...).sort(0)  (nerfBlock.scala:71)
      u.assign(u_rand + b * NI, u_rand + (b + 1) * NI);
      std::sort(u.begin(), u.end());
    }
    sample_pdf_ray(N, NI, w + b * N, z + b * N, u.data(), cdf ? cdf + b * (N - 1) : nullptr,
                   ind ? ind + b * NI : nullptr, z_samples ? z_samples + b * NI : nullptr,
                   z_fine + b * (N + NI));
  }
}

void orc_mlp_forward(const orc_config* c, const float* net_params, int32_t is_coarse, int32_t training,
                     int64_t B, int32_t S, const float* rays_o, const float* rays_d, const float* viewdir,
                     const float* z, float* sigma, float* craw) {
  const NetLayout L = make_layout(c);
  const bool want_rgb = !(is_coarse && !training);  // nerfCoreBlock.scala:109
  net_forward(c, L, net_params, want_rgb, B, S, rays_o, rays_d, viewdir, z, sigma, craw, nullptr);
}

// The same network evaluated in the SUMMATION ORDER of the fp32 MFMA kernels (nerf_amd/csrc/mlp_fwd.hip,
// mlp_common.hip.h:9-17), so that a test can hold the kernel to (near-)bitwise agreement instead of the loose bar a
// different GEMM order forces: every Linear output is  acc = bias;  for kt, q, j, h: k = 32 kt + 8 q + 4 h + j:
// acc = fmaf(w[k], x[k], acc)  (v_mfma_f32_32x32x2_f32 == an fmaf chain over its two k values, bitwise); the density
// row and the colour output rows are two chains, one per lane half h (k = 32 t + 8 q + 4 h + j), added at the end;
// the view-direction branch is un-fused mul/add in natural order plus bf (pack.hip: dir_branch_kernel).  Does not
// follow the reference more closely than net_forward -- it follows the KERNEL; with posL = dirL = 0 (no sin/cos) the
// two must agree bit for bit.
static float korder_dot(const float* w, const float* x, int K, float acc) {
  for (int kt = 0; kt * 32 < K; ++kt)
    for (int q = 0; q < 4; ++q)
      for (int j = 0; j < 4; ++j)
        for (int h = 0; h < 2; ++h) {
          const int k = 32 * kt + 8 * q + 4 * h + j;
          if (k < K) acc = fmaf(w[k], x[k], acc);
        }
  return acc;
}
static float khalf_dot(const float* w, const float* x, int K, int h) {   // one lane half of a VALU dot product
  float part = 0.f;
  for (int t = 0; t * 32 < K; ++t)
    for (int q = 0; q < 4; ++q)
      for (int j = 0; j < 4; ++j) {
        const int k = 32 * t + 8 * q + 4 * h + j;
        if (k < K) part = fmaf(x[k], w[k], part);
      }
  return part;
}

void orc_mlp_forward_korder(const orc_config* c, const float* P, int32_t want_rgb, int64_t B, int32_t S,
                            const float* rays_o, const float* rays_d, const float* viewdir, const float* z,
                            float* sigma, float* craw) {
  const NetLayout L = make_layout(c);
  const int E = L.E, W = L.W, D = L.D, H2 = L.H2, Ed = L.Ed;
  const int64_t M = B * S;
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    const int64_t b = m / S;
    float pos[3];
    for (int k = 0; k < 3; ++k) pos[k] = rays_o[b * 3 + k] + rays_d[b * 3 + k] * z[m];
    std::vector<float> enc(E), cur(W + E), nxt(W + E);
    position_code_row(pos, c->pos_l, enc.data());
    std::copy(enc.begin(), enc.end(), cur.begin());
    int cur_w = E;
    for (int l = 0; l < D; ++l) {
      const float* Wm = P + L.w_off[l];
      for (int i = 0; i < W; ++i) {
        // activations first (W is a multiple of 32, so the concatenated index walks the kernel's chunks in order),
        // then the two encoding chunks of a skip layer
        const float v = korder_dot(Wm + (int64_t)i * cur_w, cur.data(), cur_w, P[L.b_off[l] + i]);
        nxt[i] = v > 0.f ? v : 0.f;
      }
      cur_w = W;
      if (is_skip(c, l)) { std::copy(enc.begin(), enc.end(), nxt.begin() + W); cur_w = W + E; }
      std::swap(cur, nxt);
    }
    // (the kernels support skips in [0, D-2] only, so cur_w == W here)
    if ((m % S) == S - 1) sigma[m] = 0.f;
    else sigma[m] = (khalf_dot(P + L.ws_off, cur.data(), W, 0) + khalf_dot(P + L.ws_off, cur.data(), W, 1)) + P[L.bs_off];
    if (!want_rgb || !craw) continue;
    if (L.mode == 1) {
      std::vector<float> encd(Ed), hc(H2);
      position_code_row(viewdir + b * 3, c->dir_l, encd.data());
      for (int j = 0; j < H2; ++j) {
        float dv = P[L.bd_off + j];
        for (int e = 0; e < Ed; ++e) dv = dv + encd[e] * P[L.wd_off + (int64_t)j * Ed + e];   // un-fused (-ffp-contract=off)
        dv = dv + P[L.bf_off + j];
        const float v = korder_dot(P + L.wf_off + (int64_t)j * W, cur.data(), W, dv);
        hc[j] = v > 0.f ? v : 0.f;
      }
      for (int k = 0; k < 3; ++k) {
        const float* w3 = P + L.w3_off + (int64_t)k * H2;
        craw[m * 3 + k] = (khalf_dot(w3, hc.data(), H2, 0) + P[L.b3_off + k]) + khalf_dot(w3, hc.data(), H2, 1);
      }
    } else if (L.mode == 2) {
      float Y[9], q[27];
      sh2_row(viewdir + b * 3, Y);
      for (int r = 0; r < 27; ++r) q[r] = korder_dot(P + L.wf_off + (int64_t)r * W, cur.data(), W, P[L.bf_off + r]);
      for (int k = 0; k < 3; ++k) {
        float part[2] = {0.f, 0.f};
        for (int h = 0; h < 2; ++h)
          for (int r = 0; r < 16; ++r) {   // accumulator register order of tile 0: row = (r&3) + 8 (r>>2) + 4 h
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < 27 && row % 3 == k) part[h] = part[h] + q[row] * Y[row / 3];
          }
        craw[m * 3 + k] = part[0] + part[1];
      }
    } else {
      for (int k = 0; k < 3; ++k) craw[m * 3 + k] = korder_dot(P + L.wf_off + (int64_t)k * W, cur.data(), W, P[L.bf_off + k]);
    }
  }
}

}  // extern "C"

namespace {

struct FwdState {
  int64_t B = 0;
  int N = 0, S = 0;
  bool hier = false;
  std::vector<float> z_c, w_c, sig_c, craw_c, z_f, w_f, sig_f, craw_f, rgb_f, rgb_c, depth, acc;
  NetCache cache_c, cache_f;
};

// nerfBlock.forwardWithCoarse / forwardWithoutCoarse (nerfBlock.scala:108-127)
void forward_impl(const orc_config* c, const float* params, bool training, int64_t B, const float* rays_o,
                  const float* rays_d, const float* bounds, const float* viewdir, orc_io* io, FwdState& st,
                  bool keep_cache) {
  const NetLayout L = make_layout(c);
  const int N = c->n_samples, NI = c->n_importance;
  const bool hier = c->use_hierarchical != 0;
  const int S = hier ? N + NI : N;
  st.B = B; st.N = N; st.S = S; st.hier = hier;
  const float* Pc = params;                          // coarse first (nerfBlock.scala:16-19)
  const float* Pf = hier ? params + L.count : params;
  st.z_c.resize((size_t)B * N);
  orc_stratified(c, B, bounds, io ? io->t_rand : nullptr, st.z_c.data());
  st.rgb_f.assign((size_t)B * 3, 0.f);
  st.depth.assign(B, 0.f);
  st.acc.assign(B, 0.f);
  const bool use_noise = training && c->raw_noise_std > 0.f;
  if (hier) {
    st.sig_c.resize((size_t)B * N);
    st.craw_c.resize((size_t)B * N * 3);
    st.w_c.resize((size_t)B * N);
    st.rgb_c.assign((size_t)B * 3, 0.f);
    net_forward(c, L, Pc, training, B, N, rays_o, rays_d, viewdir, st.z_c.data(), st.sig_c.data(),
                training ? st.craw_c.data() : nullptr, keep_cache ? &st.cache_c : nullptr);
    orc_composite(c, B, N, st.sig_c.data(), (use_noise && io) ? io->noise_c : nullptr,
                  training ? st.craw_c.data() : nullptr, st.z_c.data(), st.w_c.data(),
                  training ? st.rgb_c.data() : nullptr, nullptr, nullptr);
    st.z_f.resize((size_t)B * S);
    orc_sample_pdf(c, B, st.w_c.data(), st.z_c.data(), io ? io->u_rand : nullptr, io ? io->cdf : nullptr,
                   io ? io->ind : nullptr, io ? io->z_samples : nullptr, st.z_f.data());
    if (io && io->z_fine_in) memcpy(st.z_f.data(), io->z_fine_in, sizeof(float) * (size_t)B * S);   // test hook, nerf_oracle.h
  } else {
    st.z_f = st.z_c;
  }
  st.sig_f.resize((size_t)B * S);
  st.craw_f.resize((size_t)B * S * 3);
  st.w_f.resize((size_t)B * S);
  net_forward(c, L, Pf, true, B, S, rays_o, rays_d, viewdir, st.z_f.data(), st.sig_f.data(), st.craw_f.data(),
              keep_cache ? &st.cache_f : nullptr);
  orc_composite(c, B, S, st.sig_f.data(), (use_noise && io) ? io->noise_f : nullptr, st.craw_f.data(),
                st.z_f.data(), st.w_f.data(), st.rgb_f.data(), st.depth.data(), st.acc.data());
  if (io) {
    auto cp = [](float* dst, const std::vector<float>& src) {
      if (dst && !src.empty()) memcpy(dst, src.data(), src.size() * sizeof(float));
    };
    cp(io->rgb_fine, st.rgb_f);
    if (training && hier) cp(io->rgb_coarse, st.rgb_c);
    cp(io->depth, st.depth);
    cp(io->acc, st.acc);
    cp(io->z_coarse, st.z_c);
    if (hier) cp(io->w_coarse, st.w_c);
    cp(io->z_fine, st.z_f);
    cp(io->w_fine, st.w_f);
    if (hier) cp(io->sigma_c, st.sig_c);
    cp(io->sigma_f, st.sig_f);
    if (hier && training) cp(io->craw_c, st.craw_c);
    cp(io->craw_f, st.craw_f);
  }
}

}  // namespace

extern "C" {

void orc_forward(const orc_config* c, const float* params, int32_t training, int64_t B, const float* rays_o,
                 const float* rays_d, const float* bounds, const float* viewdir, orc_io* io) {
  FwdState st;
  forward_impl(c, params, training != 0, B, rays_o, rays_d, bounds, viewdir, io, st, false);
}

void orc_train_grads(const orc_config* c, const float* params, int64_t B, const float* rays_o,
                     const float* rays_d, const float* bounds, const float* viewdir, const float* label,
                     orc_io* io, float* loss_fine, float* loss_coarse, float* grads) {
  const NetLayout L = make_layout(c);
  FwdState st;
  forward_impl(c, params, true, B, rays_o, rays_d, bounds, viewdir, io, st, true);
  const bool hier = st.hier;
  const int N = st.N, S = st.S;
  const int64_t total = L.count * (hier ? 2 : 1);
  memset(grads, 0, sizeof(float) * total);
  const bool use_noise = c->raw_noise_std > 0.f && io;
  // Loss.l2Loss("L2Loss", 1): mean over B*3 of (label - pred)^2 (runNerf.scala:64,72-73)
  auto mse_and_grad = [&](const std::vector<float>& pred, std::vector<float>& d) {
    double s = 0;
    d.resize((size_t)B * 3);
    const float scale = 2.0f / (float)(B * 3);
    for (int64_t i = 0; i < B * 3; ++i) {
      const float diff = label[i] - pred[i];
      s += (double)(diff * diff);
      d[i] = -diff * scale;
    }
    return (float)(s / (double)(B * 3));
  };
  std::vector<float> d_rgb, d_sig, d_craw;
  const float lf = mse_and_grad(st.rgb_f, d_rgb);
  if (loss_fine) *loss_fine = lf;
  d_sig.resize((size_t)B * S);
  d_craw.resize((size_t)B * S * 3);
  orc_composite_bwd(c, B, S, st.sig_f.data(), use_noise ? io->noise_f : nullptr, st.craw_f.data(),
                    st.z_f.data(), st.w_f.data(), d_rgb.data(), d_sig.data(), d_craw.data());
  net_backward(c, L, hier ? params + L.count : params, B, S, st.cache_f, d_sig.data(), d_craw.data(),
               hier ? grads + L.count : grads);
  if (hier) {
    const float lc = mse_and_grad(st.rgb_c, d_rgb);
    if (loss_coarse) *loss_coarse = lc;
    d_sig.resize((size_t)B * N);
    d_craw.resize((size_t)B * N * 3);
    orc_composite_bwd(c, B, N, st.sig_c.data(), use_noise ? io->noise_c : nullptr, st.craw_c.data(),
                      st.z_c.data(), st.w_c.data(), d_rgb.data(), d_sig.data(), d_craw.data());
    net_backward(c, L, params, B, N, st.cache_c, d_sig.data(), d_craw.data(), grads);
  } else if (loss_coarse) {
    *loss_coarse = 0.f;
  }
}

// Tracker.factor(): lr(t) = base * factor^t, factor stored as float (runNerf.scala:65); DJL floors the
// rate at its default minimum 1e-8.
float orc_lr(const orc_config* c, int64_t t) {
  const float factor = (float)std::pow(0.1, 1.0 / ((double)c->lrate_decay * 1000.0));
  const float base = c->lrate;
  double v = (double)base * std::pow((double)factor, (double)t);
  if (v < 1e-8) v = 1e-8;
  return (float)v;
}

// DJL Adam (third-party, recalled): bias correction folded into the step size,
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; w -= lr' m / (sqrt(v) + eps), lr' = lr sqrt(1-b2^t)/(1-b1^t)
void orc_adam(const orc_config* c, int64_t n, float* p, const float* g, float* m, float* v, int64_t t,
              float grad_scale) {
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
  const double coef1 = 1.0 - std::pow((double)b1, (double)t);
  const double coef2 = 1.0 - std::pow((double)b2, (double)t);
  const float lr = orc_lr(c, t);
  const float lrp = (float)((double)lr * std::sqrt(coef2) / coef1);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float gi = g[i] * grad_scale;
    m[i] = m[i] * b1 + gi * (1.0f - b1);
    v[i] = v[i] * b2 + gi * gi * (1.0f - b2);
    p[i] = p[i] - (m[i] * lrp) / (sqrtf(v[i]) + eps);
  }
}

// XavierInitializer(GAUSSIAN, IN, 2): N(0, 2/fan_in); biases zero.
void orc_init_params(const orc_config* c, uint64_t seed, float* params) {
  const NetLayout L = make_layout(c);
  const int nets = c->use_hierarchical ? 2 : 1;
  SplitMix rng{seed};
  const int feat_w = L.W + (is_skip(c, L.D - 1) ? L.E : 0);
  for (int n = 0; n < nets; ++n) {
    float* P = params + n * L.count;
    memset(P, 0, sizeof(float) * L.count);
    auto fill = [&](int64_t off, int64_t rows, int fan_in) {
      const double sd = std::sqrt(2.0 / fan_in);
      for (int64_t i = 0; i < rows * fan_in; ++i) P[off + i] = (float)(rng.normal() * sd);
    };
    for (int l = 0; l < L.D; ++l) fill(L.w_off[l], L.W, L.fan_in[l]);
    fill(L.ws_off, 1, feat_w);
    if (L.mode == 1) {
      fill(L.wf_off, L.H2, feat_w);
      fill(L.wd_off, L.H2, L.Ed);
      fill(L.w3_off, 3, L.H2);
    } else if (L.mode == 2) fill(L.wf_off, 27, feat_w);
    else fill(L.wf_off, 3, feat_w);
  }
}

void orc_gen_rays(int32_t H, int32_t Wd, float focal, const float* c2w, float near, float far,
                  int32_t use_ndc, float* rays_o, float* rays_d, float* bounds) {
  const float halfW = (float)(Wd * .5), halfH = (float)(H * .5);
  for (int j = 0; j < H; ++j)
    for (int i = 0; i < Wd; ++i) {
      const int64_t p = (int64_t)j * Wd + i;  // row-major pixel index (runNerf.scala:187-199)
      const float dir[3] = {((float)i - halfW) / focal, -(((float)j - halfH) / focal), -1.0f};
      float d[3], o[3];
      for (int k = 0; k < 3; ++k) {  // dirs . c2w[:3,:3]^T
        d[k] = dir[0] * c2w[k * 4 + 0] + dir[1] * c2w[k * 4 + 1] + dir[2] * c2w[k * 4 + 2];
        o[k] = c2w[k * 4 + 3];
      }
      if (use_ndc) {  // llff.scala:112-130 with near = 1
        const float nr = 1.0f;
        const float t = (o[2] + nr) / d[2];
        const float o0 = o[0] - t * d[0], o1 = o[1] - t * d[1], o2 = o[2] - t * d[2];
        const float sx = -1.0f / ((float)Wd / (2.0f * focal)), sy = -1.0f / ((float)H / (2.0f * focal));
        const float n0 = o0 / o2 * sx, n1 = o1 / o2 * sy, n2 = (2.0f * nr) / o2 + 1.0f;
        const float e0 = (d[0] / d[2] - o0 / o2) * sx, e1 = (d[1] / d[2] - o1 / o2) * sy;
        const float e2 = (-2.0f * nr) / o2;
        o[0] = n0; o[1] = n1; o[2] = n2;
        d[0] = e0; d[1] = e1; d[2] = e2;
      }
      const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      for (int k = 0; k < 3; ++k) {
        rays_o[p * 3 + k] = o[k];
        rays_d[p * 3 + k] = d[k] / nrm;
      }
      bounds[p * 2] = use_ndc ? 0.f : nrm * near;  // getDataSet.scala:112
      bounds[p * 2 + 1] = use_ndc ? nrm : nrm * far;
    }
}

void orc_to_u8(const float* rgb, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) {
    float v = rgb[i] * 255.0f;
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);  // build definition: clamp, then truncate
    out[i] = (uint8_t)v;
  }
}

}  // extern "C"
