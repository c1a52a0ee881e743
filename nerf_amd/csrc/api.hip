// api.hip -- host side of libnerfhip.so: the C ABI of include/nerfhip.h, orchestration of the hot path
// (nerfBlock.forwardWithCoarse / forwardWithoutCoarse, nerfBlock.scala:108-127) and the train step
// (runNerf.scala:94-106).  Plain HIP runtime; no torch, no Python.  Fails loudly: every HIP error is
// turned into a NERFHIP_E_* code with text; there is NO CPU fallback anywhere in this library.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <queue>
#include <vector>

#include "launchers.h"
#include "rng.hip.h"

using namespace nh;

// Scratch of nerfhip_render_dev_static (the entry point that allocates nothing): the per-call intermediates of a
// render -- depths, raw densities, weights, raw colours, per-ray direction features -- live here instead of in
// hipMalloc'ed workspace.  One copy per GPU, held until the process ends.  In HOST code this name is only a stand-in
// for the symbol: its device address is taken once per handle with hipGetSymbolAddress (nerfhip_create) and only that
// pointer is ever passed to kernels.
#define NERFHIP_STATIC_SCRATCH_BYTES (64u << 20)
static __device__ __align__(256) unsigned char scratch[NERFHIP_STATIC_SCRATCH_BYTES];

namespace {
// text of the last failed nerfhip_create of THIS thread (a failed create has no handle to carry it)
thread_local std::string g_create_error;

struct Net {
  NetLayout L{};
  int HT = 1, WT = 1;
  int64_t param_off = 0;  // offset of this network in the flat parameter buffer
  std::vector<ChunkDesc> fwd_desc, bwd_desc;
  std::vector<uint32_t> off_fwd, off_bwd;
  int n_fwd_trunk = 0;
  ChunkDesc* d_fwd_desc = nullptr;
  ChunkDesc* d_bwd_desc = nullptr;
  uint32_t* d_off_fwd = nullptr;
  uint32_t* d_off_bwd = nullptr;
  float* d_wpack_fwd = nullptr;
  float* d_wpack_bwd = nullptr;
  float* d_vecpack = nullptr;
  // bf16 path (cfg.dtype == NERFHIP_BF16): fragment streams (bf16_common.hip.h) and fp32 vectors
  std::vector<FragDesc> bf_full, bf_sigma;      // forward stream with the colour head / with the density tile only
  FragDesc* d_bf_full_desc = nullptr;
  FragDesc* d_bf_sigma_desc = nullptr;
  void* d_bf_full = nullptr;
  void* d_bf_sigma = nullptr;
  // the same two streams in the fragment order of the 16x16x32 inference kernel (mlp_fwd_bf16_t16.hip)
  std::vector<FragDesc> bf_full16, bf_sigma16;
  FragDesc* d_bf_full16_desc = nullptr;
  FragDesc* d_bf_sigma16_desc = nullptr;
  void* d_bf_full16 = nullptr;
  void* d_bf_sigma16 = nullptr;
  float* d_bvec = nullptr;
  std::vector<FragDesc> bf_bwd;                 // backward stream (transposed matrices)
  FragDesc* d_bf_bwd_desc = nullptr;
  void* d_bf_bwd = nullptr;
  // split-f16 path (cfg.dtype == NERFHIP_F16X3): forward streams of (hi, lo) fragment pairs (f16x3_common.hip.h)
  std::vector<FragDesc> x3_full, x3_sigma, x3_bwd;   // x3_bwd: transposed matrices, for the backward kernel
  // the two forward pair streams in the fragment order of the 16x16x32 inference kernel (mlp_fwd_f16x3_t16.hip)
  std::vector<FragDesc> x3_full16, x3_sigma16;
  FragDesc* d_x3_full16_desc = nullptr;
  FragDesc* d_x3_sigma16_desc = nullptr;
  void* d_x3_full16 = nullptr;
  void* d_x3_sigma16 = nullptr;
  FragDesc* d_x3_full_desc = nullptr;
  FragDesc* d_x3_sigma_desc = nullptr;
  FragDesc* d_x3_bwd_desc = nullptr;
  void* d_x3_full = nullptr;
  void* d_x3_sigma = nullptr;
  void* d_x3_bwd = nullptr;
  BfStore bst{};                                // bf16 training store (allocated on first training call)
  float* d_dirfeat = nullptr;  // [max_rays][max(32*HT, 12)]
  float* d_encdirT = nullptr;  // training, mode 1: [32][Bpad]
  float* d_ddirT = nullptr;    // training, mode 1: d(dirvec)^T [32*HT][Bpad]
  ActStore st{};               // training activation store (allocated on first training call)
  float* d_dsigma = nullptr;   // [max_rays][S]
  float* d_dcraw = nullptr;    // [max_rays][S][3]
};

enum ProfClass { P_MLP_FWD = 0, P_MLP_BWD, P_WGRAD, P_COMPOSITE, P_SAMPLING, P_OTHER, P_COMM, P_NCLASS };
struct ProfRec { int cls; hipEvent_t a, b; };
}  // namespace

struct nerfhip {
  nerfhip_config cfg{};
  int nnets = 1;
  Net nets[2];
  int64_t nparams = 0;
  int N = 0, NI = 0, S = 0;
  int64_t R = 0;  // max rays per call
  float* d_params = nullptr;
  float* d_grads = nullptr;
  float* d_m = nullptr;
  float* d_v = nullptr;
  int64_t opt_step = 0;
  uint32_t rng_step = 0, last_rng_step = 0;
  bool grads_ready = false;
  bool packed = false;
  bool x3_infer_dirty = true;   // dtype f16x3: the inference-order pair streams lag behind the parameters (packed on demand)
  hipStream_t own_stream = nullptr, stream = nullptr;
  float* d_pose_rays = nullptr;   // nerfhip_render_pose_u8: rays_o | rays_d | bounds of one frame
  int64_t pose_rays_cap = 0;
  // workspace (device)
  float *w_o = nullptr, *w_d = nullptr, *w_b = nullptr, *w_v = nullptr, *w_label = nullptr;
  float *w_rgb_f = nullptr, *w_rgb_c = nullptr, *w_depth = nullptr, *w_acc = nullptr, *w_drgb = nullptr;
  float *w_zc = nullptr, *w_sigc = nullptr, *w_wc = nullptr, *w_crawc = nullptr;
  float *w_zf = nullptr, *w_sigf = nullptr, *w_wf = nullptr, *w_crawf = nullptr;
  float *w_trand = nullptr, *w_urand = nullptr, *w_noisec = nullptr, *w_noisef = nullptr;
  uint8_t* w_u8 = nullptr;
  double* d_loss = nullptr;  // [3]: fine and coarse loss sums; [2] = 8 bytes of scratch for dp_broadcast_state's update count
  bool train_ws = false;
  nh::WgradItem* d_items = nullptr;
  nh::WgradBfItem* d_bitems = nullptr;   // bf16 path
  std::vector<nh::WgradBfItem> bitems;
  uint32_t* d_x3flags = nullptr;      // dtype f16x3 range guard: [net] = flat index + 1 of a weight with |64 w| >= 65504 (pack.hip),
                                      // [2] = a forward kernel wrote a non-finite output (mlp_fwd_f16x3.hip); see check_x3_range
  float* d_amax = nullptr;            // dtype f16x3: [2][kMaxD + 2] largest gradient elements of the step (mlp_bwd_f16x3.hip)
  int32_t* d_items_first = nullptr;   // [n workgroups + 1] first work entry of every persistent workgroup
  std::vector<int32_t> items_first;
  std::vector<nh::WgradItem> items;
  int64_t items_nr = -1; // ray count the work list in d_items was built for
  double items_flop = 0;
  int n_cu = 256;
  // data parallelism (dp_rccl.inc): ncclComm_t of this replica, null = single device
  unsigned char* static_scratch = nullptr;   // device address of `scratch` on cfg.device (hipGetSymbolAddress)
  void* dp_comm = nullptr;
  int dp_rank = 0, dp_world = 1;
  // profiling
  bool prof_on = false;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  double flop_fwd = 0, flop_bwd = 0, flop_wgrad = 0;
  int fwd_launches = 0;
  std::string err;
};

namespace {

int fail(nerfhip* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}
#define HIPCHK(h, expr)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail(h, e_ == hipErrorOutOfMemory ? NERFHIP_E_NOMEM : NERFHIP_E_HIP,                \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                            \
  } while (0)

bool is_skip(const nerfhip_config& c, int i) {
  for (int k = 0; k < c.n_skips; ++k)
    if (c.skips[k] == i) return true;
  return false;
}

NetLayout make_layout(const nerfhip_config& c) {
  NetLayout L{};
  L.D = c.D; L.W = c.W; L.E = 3 * (1 + 2 * c.pos_l); L.Ed = 3 * (1 + 2 * c.dir_l); L.H2 = c.W / 2;
  L.mode = c.use_dir ? (c.use_sh ? 2 : 1) : 0;
  L.head_rows = L.mode == 1 ? L.H2 : (L.mode == 2 ? 27 : 3);
  int64_t off = 0;
  for (int l = 0; l < L.D; ++l) {
    L.fan_in[l] = (l == 0) ? L.E : (L.W + (is_skip(c, l - 1) ? L.E : 0));
    if (is_skip(c, l)) L.skip_mask |= 1u << l;
    L.w_off[l] = off; off += (int64_t)L.W * L.fan_in[l];
    L.b_off[l] = off; off += L.W;
  }
  L.ws_off = off; off += L.W;
  L.bs_off = off; off += 1;
  L.wf_off = off; off += (int64_t)L.head_rows * L.W;
  L.bf_off = off; off += L.head_rows;
  if (L.mode == 1) {
    L.wd_off = off; off += (int64_t)L.H2 * L.Ed;
    L.bd_off = off; off += L.H2;
    L.wc_off = off; off += 3 * L.H2;
    L.bc_off = off; off += 3;
  }
  L.count = off;
  return L;
}

void build_chunks(Net& n) {
  const NetLayout& L = n.L;
  const int W = L.W, WT = n.WT, HT = n.HT;
  uint32_t off = 0;
  auto push = [&](std::vector<ChunkDesc>& v, std::vector<uint32_t>& offs, ChunkDesc d) {
    d.dst_off = off;
    offs.push_back(off);
    off += 32u * (uint32_t)d.R;
    v.push_back(d);
  };
  // forward stream
  for (int l = 0; l < L.D; ++l) {
    const bool uses_act = l > 0;
    const bool uses_enc = l == 0 || ((L.skip_mask >> (l - 1)) & 1u);
    if (uses_act)
      for (int kt = 0; kt < WT; ++kt)
        push(n.fwd_desc, n.off_fwd, ChunkDesc{L.w_off[l], L.fan_in[l], 0, 32 * kt, 32, W, W, 0, 0});
    if (uses_enc)
      for (int kt = 0; kt < 2; ++kt) {
        const int kv = std::min(32, L.E - 32 * kt);
        push(n.fwd_desc, n.off_fwd,
             ChunkDesc{L.w_off[l], L.fan_in[l], 0, kv > 0 ? (uses_act ? W : 0) + 32 * kt : -1, std::max(kv, 0), W, W, 0, 0});
      }
  }
  n.n_fwd_trunk = (int)n.fwd_desc.size();
  for (int kt = 0; kt < WT; ++kt)
    push(n.fwd_desc, n.off_fwd, ChunkDesc{L.wf_off, W, 0, 32 * kt, 32, L.head_rows, 32 * HT, 0, 0});
  n.off_fwd.push_back(off);
  // backward stream (transposed matrices)
  off = 0;
  for (int kt = 0; kt < HT; ++kt)
    push(n.bwd_desc, n.off_bwd, ChunkDesc{L.wf_off, W, 1, 32 * kt, std::max(0, std::min(32, L.head_rows - 32 * kt)), W, W, 0, 0});
  for (int l = L.D - 1; l >= 1; --l)
    for (int kt = 0; kt < WT; ++kt)
      push(n.bwd_desc, n.off_bwd, ChunkDesc{L.w_off[l], L.fan_in[l], 1, 32 * kt, 32, W, W, 0, 0});
  n.off_bwd.push_back(off);
}

// bf16 forward streams: fragments in consumption order, every layer part padded to a whole 32-fragment block
// (pad_to = 32); the split-f16 kernel stores a (hi, lo) PAIR per descriptor, so its parts are padded to 16
void build_fwd_streams(const Net& n, int pad_to, std::vector<FragDesc>& out_full, std::vector<FragDesc>& out_sigma) {
  const NetLayout& L = n.L;
  const int W = L.W, WT = n.WT;
  const int HT = L.mode == 1 ? std::max(1, WT / 2) : 1;   // colour-hidden tiles (kernel template)
  auto pad = [pad_to](std::vector<FragDesc>& v) {
    while (v.size() % pad_to) v.push_back(FragDesc{-1, 0, 0, 0, 0, 0, 0, 0});
  };
  auto ksteps = [&](std::vector<FragDesc>& v, int64_t off, int ld, int row0, int nrows, int col_base, int kfeat) {
    for (int kt = 0; kt * 32 < kfeat; ++kt)
      for (int s = 0; s < 2; ++s)
        v.push_back(FragDesc{off, ld, row0, nrows, col_base + 32 * kt, std::max(0, std::min(32, kfeat - 32 * kt)), s, 0});
  };
  auto trunk = [&](std::vector<FragDesc>& v) {
    for (int l = 0; l < L.D; ++l) {
      const bool uses_act = l > 0;
      const bool uses_enc = l == 0 || ((L.skip_mask >> (l - 1)) & 1u);
      for (int ot = 0; ot < WT; ++ot) {
        if (uses_act) ksteps(v, L.w_off[l], L.fan_in[l], 32 * ot, W, 0, W);
        if (uses_enc) {   // always 2 k-tiles (64 padded encoding rows)
          const int c0 = uses_act ? W : 0;
          for (int kt = 0; kt < 2; ++kt)
            for (int s = 0; s < 2; ++s)
              v.push_back(FragDesc{L.w_off[l], L.fan_in[l], 32 * ot, W, c0 + 32 * kt, std::max(0, std::min(32, L.E - 32 * kt)), s, 0});
        }
      }
      pad(v);
    }
  };
  out_full.clear(); out_sigma.clear();
  trunk(out_full);
  for (int ot = 0; ot < HT; ++ot) ksteps(out_full, L.wf_off, W, 32 * ot, L.head_rows, 0, W);
  ksteps(out_full, L.ws_off, W, 0, 1, 0, W);   // density tile: row 0
  pad(out_full);
  if (L.mode == 1) {
    for (int kt = 0; kt < HT; ++kt)
      for (int s = 0; s < 2; ++s)
        out_full.push_back(FragDesc{L.wc_off, L.H2, 0, 3, 32 * kt, std::max(0, std::min(32, L.H2 - 32 * kt)), s, 0});
    pad(out_full);
  }
  trunk(out_sigma);
  ksteps(out_sigma, L.ws_off, W, 0, 1, 0, W);
  pad(out_sigma);

}

// backward stream (mlp_bwd_bf16.hip, mlp_bwd_f16x3.hip): element (row = input feature i, k = output feature) = M[k][i];
// every part padded to pad_to descriptors (one ring block: 32 fragments, or 16 (hi, lo) pairs)
void build_bwd_stream(const Net& n, int pad_to, std::vector<FragDesc>& out) {
  const NetLayout& L = n.L;
  const int W = L.W, WT = n.WT;
  const int HT = L.mode == 1 ? std::max(1, WT / 2) : 1;
  auto pad = [pad_to](std::vector<FragDesc>& v) {
    while (v.size() % pad_to) v.push_back(FragDesc{-1, 0, 0, 0, 0, 0, 0, 0});
  };
  auto tfrag = [&](int64_t off, int ld, int row0, int nrows, int col0, int kvalid, int sidx) {
    out.push_back(FragDesc{off, ld, row0, nrows, col0, std::max(0, std::min(32, kvalid)), sidx, 1});
  };
  out.clear();
  if (L.mode == 1) {   // Wc^T: rows = hidden features, k = colour channel (3)
    for (int ot = 0; ot < HT; ++ot) tfrag(L.wc_off, L.H2, 32 * ot, L.H2, 0, 3, 0);
    pad(out);
  }
  for (int ot = 0; ot < WT; ++ot) {   // head: Wf^T (k = head row) then the density k-step (k = 0: w_sigma)
    for (int kt = 0; kt < HT; ++kt)
      for (int sidx = 0; sidx < 2; ++sidx) tfrag(L.wf_off, W, 32 * ot, W, 32 * kt, L.head_rows - 32 * kt, sidx);
    tfrag(L.ws_off, W, 32 * ot, W, 0, 1, 0);
  }
  pad(out);
  for (int l = L.D - 1; l >= 1; --l) {   // W_l^T restricted to the W activation inputs (the encoding has no gradient)
    for (int ot = 0; ot < WT; ++ot)
      for (int kt = 0; kt < WT; ++kt)
        for (int sidx = 0; sidx < 2; ++sidx) tfrag(L.w_off[l], L.fan_in[l], 32 * ot, W, 32 * kt, 32, sidx);
    pad(out);
  }
}

// forward streams of the 16x16x32 inference kernel (mlp_fwd_bf16_t16.hip): a fragment = 16 output rows x one k-step of 32
// features (FragDesc::transposed == 2), fragments in consumption order (output tile outer, k-step inner; a skip layer's
// encoding k-steps behind its activation k-steps), every part padded to a whole ring block
void build_fwd_streams_t16(const Net& n, int pad_to, std::vector<FragDesc>& out_full, std::vector<FragDesc>& out_sigma) {
  const NetLayout& L = n.L;
  const int W = L.W, WT = n.WT;
  const int HT = L.mode == 1 ? std::max(1, WT / 2) : 1;   // colour-hidden tiles of 32 rows (kernel template)
  auto pad = [pad_to](std::vector<FragDesc>& v) {   // 32 fragments, or 16 (hi, lo) pairs: one ring block
    while (v.size() % pad_to) v.push_back(FragDesc{-1, 0, 0, 0, 0, 0, 0, 2});
  };
  auto ksteps = [&](std::vector<FragDesc>& v, int64_t off, int ld, int row0, int nrows, int col_base, int kfeat, int nks) {
    for (int ks = 0; ks < nks; ++ks)
      v.push_back(FragDesc{off, ld, row0, nrows, col_base + 32 * ks, std::max(0, std::min(32, kfeat - 32 * ks)), 0, 2});
  };
  auto trunk = [&](std::vector<FragDesc>& v) {
    for (int l = 0; l < L.D; ++l) {
      const bool uses_act = l > 0;
      const bool uses_enc = l == 0 || ((L.skip_mask >> (l - 1)) & 1u);
      for (int ot = 0; ot < 2 * WT; ++ot) {
        if (uses_act) ksteps(v, L.w_off[l], L.fan_in[l], 16 * ot, W, 0, W, WT);
        if (uses_enc) ksteps(v, L.w_off[l], L.fan_in[l], 16 * ot, W, uses_act ? W : 0, L.E, 2);   // 64 padded encoding rows
      }
      pad(v);
    }
  };
  out_full.clear(); out_sigma.clear();
  trunk(out_full);
  for (int ot = 0; ot < 2 * HT; ++ot) ksteps(out_full, L.wf_off, W, 16 * ot, L.head_rows, 0, W, WT);
  ksteps(out_full, L.ws_off, W, 0, 1, 0, W, WT);   // density tile: row 0
  pad(out_full);
  if (L.mode == 1) {
    ksteps(out_full, L.wc_off, L.H2, 0, 3, 0, L.H2, HT);
    pad(out_full);
  }
  trunk(out_sigma);
  ksteps(out_sigma, L.ws_off, W, 0, 1, 0, W, WT);
  pad(out_sigma);
}

// Inference forward of the bf16 path: 1 = mlp_fwd_bf16_t16.hip (v_mfma_f32_16x16x32_bf16), 0 = the 32x32x16 kernel the
// training forward also uses (mlp_fwd_bf16.hip).  Same-box A/B in round 3: see DESIGN.md section 5.
#ifndef NH_BF_INFER_T16
#define NH_BF_INFER_T16 1
#endif
#ifndef NH_X3_INFER_T16   // the same switch for dtype f16x3 (mlp_fwd_f16x3_t16.hip against mlp_fwd_f16x3.hip)
#define NH_X3_INFER_T16 1
#endif
void build_bf16_streams(Net& n) {
  build_fwd_streams(n, 32, n.bf_full, n.bf_sigma);
  build_fwd_streams_t16(n, 32, n.bf_full16, n.bf_sigma16);
  build_bwd_stream(n, 32, n.bf_bwd);
}

NetArgs net_args(const nerfhip* h, const Net& n) {
  NetArgs a{};
  a.params = h->d_params + n.param_off;
  a.vecpack = n.d_vecpack;
  a.wpack_fwd = n.d_wpack_fwd;
  a.wpack_bwd = n.d_wpack_bwd;
  a.off_fwd = n.d_off_fwd;
  a.off_bwd = n.d_off_bwd;
  a.n_fwd = (int)n.fwd_desc.size();
  a.n_fwd_trunk = n.n_fwd_trunk;
  a.n_bwd = (int)n.bwd_desc.size();
  a.D = n.L.D; a.W = n.L.W; a.E = n.L.E; a.mode = n.L.mode; a.HT = n.HT; a.head_rows = n.L.head_rows; a.H2 = n.L.H2;
  a.skip_mask = n.L.skip_mask;
  return a;
}

template <typename T>
hipError_t dalloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) return hipSuccess;
  return hipMalloc((void**)p, count * sizeof(T));
}

// ---- profiling helpers -------------------------------------------------------------------------
struct Scope {
  nerfhip* h; int cls; hipEvent_t a = nullptr, b = nullptr;
  Scope(nerfhip* h_, int cls_) : h(h_), cls(cls_) {
    if (!h->prof_on) return;
    auto get = [&]() { hipEvent_t e; if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); } else hipEventCreate(&e); return e; };
    a = get(); b = get();
    hipEventRecord(a, h->stream);
  }
  ~Scope() {
    if (!h->prof_on) return;
    hipEventRecord(b, h->stream);
    h->prof.push_back(ProfRec{cls, a, b});
  }
};
void prof_reset(nerfhip* h) {
  for (auto& r : h->prof) { h->ev_pool.push_back(r.a); h->ev_pool.push_back(r.b); }
  h->prof.clear();
  h->flop_fwd = h->flop_bwd = h->flop_wgrad = 0;
  h->fwd_launches = 0;
}

double net_fwd_flop_per_sample(const Net& n, bool want_rgb) {
  const NetLayout& L = n.L;
  double mac = 0;
  for (int l = 0; l < L.D; ++l) mac += (double)L.W * L.fan_in[l];
  mac += L.W;  // density head
  if (want_rgb) {
    mac += (double)L.head_rows * L.W;
    if (L.mode == 1) mac += 3.0 * L.H2;
  }
  return 2.0 * mac;
}

int repack(nerfhip* h) {
  Scope sc(h, P_OTHER);
  if (h->cfg.dtype == NERFHIP_BF16) {   // one launch for every stream and vector of every network
    RepackBf16Args a{};
    for (int i = 0; i < h->nnets; ++i) {
      Net& n = h->nets[i];
      const float* p = h->d_params + n.param_off;
      auto seg = [&](const FragDesc* d, void* dst, size_t cnt) {
        a.seg[a.nseg++] = RepackBf16Args::Seg{d, dst, p, a.nfrag, (int32_t)cnt};
        a.nfrag += (int32_t)cnt;
      };
      seg(n.d_bf_full_desc, n.d_bf_full, n.bf_full.size());
      if (!NH_BF_INFER_T16) seg(n.d_bf_sigma_desc, n.d_bf_sigma, n.bf_sigma.size());   // density-only 32x32x16 stream: inference only
      seg(n.d_bf_full16_desc, n.d_bf_full16, n.bf_full16.size());
      seg(n.d_bf_sigma16_desc, n.d_bf_sigma16, n.bf_sigma16.size());
      seg(n.d_bf_bwd_desc, n.d_bf_bwd, n.bf_bwd.size());
      a.vparams[i] = p; a.bv[i] = n.d_bvec; a.L[i] = n.L; a.HT[i] = n.L.mode == 1 ? std::max(1, n.WT / 2) : 1;
    }
    a.nnets = h->nnets;
    HIPCHK(h, launch_repack_bf16(a, h->stream));
    h->packed = true;
    return NERFHIP_OK;
  }
  for (int i = 0; i < h->nnets; ++i) {
    Net& n = h->nets[i];
    const float* p = h->d_params + n.param_off;
    if (h->cfg.dtype != NERFHIP_F16X3) {   // fp32 tile images: only the fp32 MFMA kernels read them
      HIPCHK(h, launch_pack_chunks(p, n.d_fwd_desc, (int)n.fwd_desc.size(), n.d_wpack_fwd, h->stream));
      HIPCHK(h, launch_pack_chunks(p, n.d_bwd_desc, (int)n.bwd_desc.size(), n.d_wpack_bwd, h->stream));
    }
    HIPCHK(h, launch_pack_vec(p, n.L, n.HT, n.d_vecpack, h->stream));
    if (h->cfg.dtype == NERFHIP_F16X3) {   // the split-f16 kernels' pair streams (forward, density-only, backward)
      HIPCHK(h, launch_pack_frags_f16x3(p, n.d_x3_full_desc, (int)n.x3_full.size(), n.d_x3_full, h->d_x3flags + i, h->stream));
      if (!NH_X3_INFER_T16)   // the density-only 32x32x16 stream is read by the old inference kernel only
        HIPCHK(h, launch_pack_frags_f16x3(p, n.d_x3_sigma_desc, (int)n.x3_sigma.size(), n.d_x3_sigma, nullptr, h->stream));
      HIPCHK(h, launch_pack_frags_f16x3(p, n.d_x3_bwd_desc, (int)n.x3_bwd.size(), n.d_x3_bwd, nullptr, h->stream));
      HIPCHK(h, launch_pack_bvec(p, n.L, n.L.mode == 1 ? std::max(1, n.WT / 2) : 1, n.d_bvec, 64.0f, h->stream));
      h->x3_infer_dirty = true;   // the 16x16x32-ordered streams are packed when an inference pass asks for them
    }
  }
  h->packed = true;
  return NERFHIP_OK;
}

int dir_features(nerfhip* h, Net& n, const float* viewdir, int64_t nr, bool training) {
  if (n.L.mode == 0) return NERFHIP_OK;
  Scope sc(h, P_OTHER);
  if (n.L.mode == 1)
    HIPCHK(h, launch_dir_branch(h->d_params + n.param_off, n.L, 32 * n.HT, viewdir, nr, n.d_dirfeat,
                                training ? n.d_encdirT : nullptr, (nr + 31) / 32 * 32, h->stream));
  else
    HIPCHK(h, launch_sh2(viewdir, nr, n.d_dirfeat, h->stream));
  return NERFHIP_OK;
}

int mlp_forward(nerfhip* h, Net& n, bool train_store, const float* o, const float* d, const float* z, int64_t nr, int S,
                float* sigma, float* craw, bool want_rgb) {
  Scope sc(h, P_MLP_FWD);
  if (h->cfg.dtype == NERFHIP_F16X3 && NH_X3_INFER_T16 && !train_store) {   // inference: the 16x16x32 kernel on its own pair order
    if (h->x3_infer_dirty) {   // a training loop never pays for these two streams
      for (int i = 0; i < h->nnets; ++i) {
        Net& q = h->nets[i];
        const float* p = h->d_params + q.param_off;
        HIPCHK(h, launch_pack_frags_f16x3(p, q.d_x3_full16_desc, (int)q.x3_full16.size(), q.d_x3_full16, nullptr, h->stream));
        HIPCHK(h, launch_pack_frags_f16x3(p, q.d_x3_sigma16_desc, (int)q.x3_sigma16.size(), q.d_x3_sigma16, nullptr, h->stream));
      }
      h->x3_infer_dirty = false;
    }
    HIPCHK(h, launch_mlp_fwd_f16x3_t16(net_args(h, n), want_rgb ? n.d_x3_full16 : n.d_x3_sigma16, n.d_bvec, o, d, z, n.d_dirfeat, nr * S, S,
                                       sigma, craw, want_rgb ? 1 : 0, (int)((want_rgb ? n.x3_full16.size() : n.x3_sigma16.size()) / 16),
                                       h->n_cu, h->d_x3flags + 2, h->stream));
  } else if (h->cfg.dtype == NERFHIP_F16X3) {   // training forward: the 32x32x16 kernels, filling the fp32 path's activation store
    ActStore st = n.st;
    st.Mpad = (nr * S + kTile - 1) / kTile * kTile;
    HIPCHK(h, launch_mlp_fwd_f16x3(net_args(h, n), want_rgb ? n.d_x3_full : n.d_x3_sigma, n.d_bvec, o, d, z, n.d_dirfeat, nr * S, S,
                                   sigma, craw, want_rgb ? 1 : 0, (int)((want_rgb ? n.x3_full.size() : n.x3_sigma.size()) / 16),
                                   h->n_cu, train_store ? &st : nullptr, h->d_x3flags + 2, h->stream));
  } else if (h->cfg.dtype == NERFHIP_BF16) {
    BfStore bst = n.bst;
    bst.Mpad = (nr * S + 255) / 256 * 256;
    if (NH_BF_INFER_T16 && !train_store)   // inference: the 16x16x32 kernel on its own fragment order
      HIPCHK(h, launch_mlp_fwd_bf16_t16(net_args(h, n), want_rgb ? n.d_bf_full16 : n.d_bf_sigma16, n.d_bvec, o, d, z, n.d_dirfeat,
                                        nr * S, S, sigma, craw, want_rgb ? 1 : 0,
                                        (int)((want_rgb ? n.bf_full16.size() : n.bf_sigma16.size()) / 32), h->n_cu, h->stream));
    else   // training forward (and, with NH_BF_INFER_T16 = 0, inference): the 32x32x16 kernel
      HIPCHK(h, launch_mlp_fwd_bf16(train_store ? &bst : nullptr, net_args(h, n), want_rgb ? n.d_bf_full : n.d_bf_sigma, n.d_bvec, o, d, z,
                                    n.d_dirfeat, nr * S, S, sigma, craw, want_rgb ? 1 : 0,
                                    (int)((want_rgb ? n.bf_full.size() : n.bf_sigma.size()) / 32), h->n_cu, h->stream));
  } else {
    ActStore st = n.st;
    st.Mpad = (nr * S + kTile - 1) / kTile * kTile;
    HIPCHK(h, launch_mlp_fwd_f32(train_store, net_args(h, n), o, d, z, n.d_dirfeat, nr * S, S, sigma, craw,
                                 want_rgb ? 1 : 0, st, h->stream));
  }
  h->flop_fwd += net_fwd_flop_per_sample(n, want_rgb) * (double)nr * S;
  h->fwd_launches++;
  return NERFHIP_OK;
}

struct DevIn { const float *o, *d, *b, *v; const float *t_rand, *u_rand, *noise_c, *noise_f; };

// nerfBlock.forwardInternal on device buffers; results land in the workspace.
int forward_dev(nerfhip* h, const DevIn& in, int64_t nr, bool training, bool train_store, bool own_bounds = false) {
  if (nr <= 0 || (!own_bounds && nr > h->R)) return fail(h, NERFHIP_E_INVALID, "n out of range (0 < n <= max_rays)");
  if (!h->packed) { int rc = repack(h); if (rc) return rc; }
  const nerfhip_config& c = h->cfg;
  const int N = h->N, NI = h->NI, S = h->S;
  const bool hier = c.use_hierarchical != 0;
  const uint32_t step = h->rng_step++;
  h->last_rng_step = step;
  const float nstd = (training && c.raw_noise_std > 0.f) ? c.raw_noise_std : 0.f;
  { Scope sc(h, P_SAMPLING);
    HIPCHK(h, launch_stratified(in.b, in.t_rand, nr, N, c.lin_disp, c.perturb, c.seed, step, h->w_zc, h->stream)); }
  Net& fine = h->nets[h->nnets - 1];
  if (hier) {
    Net& coarse = h->nets[0];
    if (training) { int rc = dir_features(h, coarse, in.v, nr, train_store); if (rc) return rc; }
    int rc = mlp_forward(h, coarse, train_store, in.o, in.d, h->w_zc, nr, N, h->w_sigc, h->w_crawc, training);
    if (rc) return rc;
    { Scope sc(h, P_COMPOSITE);
      HIPCHK(h, launch_composite_fwd(h->w_sigc, in.noise_c, training ? h->w_crawc : nullptr, h->w_zc, nr, N, nstd,
                                     c.white_bkgd, c.seed, step, kStreamNoiseC, h->w_wc, training ? h->w_rgb_c : nullptr,
                                     nullptr, nullptr, h->stream)); }
    { Scope sc(h, P_SAMPLING);
      HIPCHK(h, launch_sample_pdf(h->w_wc, h->w_zc, in.u_rand, nr, N, NI, c.perturb, c.seed, step, nullptr, nullptr,
                                  nullptr, h->w_zf, h->stream)); }
  }
  const float* zf = hier ? h->w_zf : h->w_zc;
  int rc = dir_features(h, fine, in.v, nr, train_store);
  if (rc) return rc;
  rc = mlp_forward(h, fine, train_store, in.o, in.d, zf, nr, S, h->w_sigf, h->w_crawf, true);
  if (rc) return rc;
  { Scope sc(h, P_COMPOSITE);
    HIPCHK(h, launch_composite_fwd(h->w_sigf, in.noise_f, h->w_crawf, zf, nr, S, nstd, c.white_bkgd, c.seed, step,
                                   kStreamNoiseF, h->w_wf, h->w_rgb_f, h->w_depth, h->w_acc, h->stream)); }
  return NERFHIP_OK;
}

int copy_out(nerfhip* h, void* dst, const void* src, size_t bytes, bool host) {
  if (!dst) return NERFHIP_OK;
  HIPCHK(h, hipMemcpyAsync(dst, src, bytes, host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream));
  return NERFHIP_OK;
}

int export_aux(nerfhip* h, nerfhip_aux* a, int64_t nr, bool training, bool host) {
  if (!a) return NERFHIP_OK;
  const int N = h->N, S = h->S;
  const bool hier = h->cfg.use_hierarchical != 0;
  const size_t f = sizeof(float);
  int rc = 0;
  rc |= copy_out(h, a->z_coarse, h->w_zc, nr * N * f, host);
  if (hier) {
    rc |= copy_out(h, a->w_coarse, h->w_wc, nr * N * f, host);
    rc |= copy_out(h, a->sigma_c, h->w_sigc, nr * N * f, host);
    if (training) rc |= copy_out(h, a->craw_c, h->w_crawc, nr * N * 3 * f, host);
  }
  rc |= copy_out(h, a->z_fine, hier ? h->w_zf : h->w_zc, nr * S * f, host);
  rc |= copy_out(h, a->w_fine, h->w_wf, nr * S * f, host);
  rc |= copy_out(h, a->sigma_f, h->w_sigf, nr * S * f, host);
  rc |= copy_out(h, a->craw_f, h->w_crawf, nr * S * 3 * f, host);
  rc |= copy_out(h, a->depth, h->w_depth, nr * f, host);
  rc |= copy_out(h, a->acc, h->w_acc, nr * f, host);
  return rc ? NERFHIP_E_HIP : NERFHIP_OK;
}

// stage host inputs of one slice into the workspace
int stage_in(nerfhip* h, const float* o, const float* d, const float* b, const float* v, const float* label,
             const nerfhip_aux* a, int64_t off, int64_t nr, DevIn* in) {
  const size_t f = sizeof(float);
  auto up = [&](float* dst, const float* src, size_t per) -> hipError_t {
    return hipMemcpyAsync(dst, src + off * per, nr * per * f, hipMemcpyHostToDevice, h->stream);
  };
  HIPCHK(h, up(h->w_o, o, 3));
  HIPCHK(h, up(h->w_d, d, 3));
  HIPCHK(h, up(h->w_b, b, 2));
  HIPCHK(h, up(h->w_v, v, 3));
  if (label) HIPCHK(h, up(h->w_label, label, 3));
  *in = DevIn{h->w_o, h->w_d, h->w_b, h->w_v, nullptr, nullptr, nullptr, nullptr};
  if (a) {
    if (a->t_rand) { HIPCHK(h, up(h->w_trand, a->t_rand, h->N)); in->t_rand = h->w_trand; }
    if (a->u_rand && h->NI > 0) { HIPCHK(h, up(h->w_urand, a->u_rand, h->NI)); in->u_rand = h->w_urand; }
    if (a->noise_c) { HIPCHK(h, up(h->w_noisec, a->noise_c, h->N)); in->noise_c = h->w_noisec; }
    if (a->noise_f) { HIPCHK(h, up(h->w_noisef, a->noise_f, h->S)); in->noise_f = h->w_noisef; }
  }
  return NERFHIP_OK;
}

nerfhip_aux aux_slice(const nerfhip_aux* a, int64_t off, int N, int S) {
  nerfhip_aux s{};
  if (!a) return s;
  auto adv = [&](float* p, int64_t per) { return p ? p + off * per : nullptr; };
  s.z_coarse = adv(a->z_coarse, N); s.w_coarse = adv(a->w_coarse, N);
  s.z_fine = adv(a->z_fine, S); s.w_fine = adv(a->w_fine, S);
  s.sigma_c = adv(a->sigma_c, N); s.sigma_f = adv(a->sigma_f, S);
  s.craw_c = adv(a->craw_c, 3 * (int64_t)N); s.craw_f = adv(a->craw_f, 3 * (int64_t)S);
  s.depth = adv(a->depth, 1); s.acc = adv(a->acc, 1);
  return s;
}

// name of the parameter tensor that holds flat index `at` of one network (DJL child order, include/nerfhip.h)
std::string param_name(const NetLayout& L, int64_t at) {
  auto in = [&](int64_t off, int64_t cnt) { return at >= off && at < off + cnt; };
  for (int l = 0; l < L.D; ++l) {
    if (in(L.w_off[l], (int64_t)L.W * L.fan_in[l])) return "trunk layer " + std::to_string(l) + " weight";
    if (in(L.b_off[l], L.W)) return "trunk layer " + std::to_string(l) + " bias";
  }
  if (in(L.ws_off, L.W)) return "density head weight";
  if (in(L.wf_off, (int64_t)L.head_rows * L.W)) return "colour head first matrix";
  if (L.mode == 1 && in(L.wd_off, (int64_t)L.H2 * L.Ed)) return "view-direction matrix";
  if (L.mode == 1 && in(L.wc_off, 3 * L.H2)) return "colour output matrix";
  return "parameter";
}

// dtype f16x3 range guard, called by every entry point that synchronises anyway (the asynchronous _dev entry points are
// covered by the next such call, e.g. nerfhip_synchronize): the split-f16 operands hold 64 w and the activations as f16
// (hi, lo) pairs, so |w| >= 1023.5 or an activation beyond 65504 has no representation.  The kernels flag both instead
// of silently producing inf; this turns a raised flag into NERFHIP_E_INVALID and clears it.  `h->stream` must be idle.
int check_x3_range(nerfhip* h) {
  if (h->cfg.dtype != NERFHIP_F16X3 || !h->d_x3flags) return NERFHIP_OK;
  uint32_t f[4] = {0, 0, 0, 0};
  HIPCHK(h, hipMemcpyAsync(f, h->d_x3flags, sizeof(f), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!(f[0] | f[1] | f[2])) return NERFHIP_OK;
  HIPCHK(h, hipMemsetAsync(h->d_x3flags, 0, sizeof(f), h->stream));
  for (int i = 0; i < h->nnets; ++i)
    if (f[i]) {
      const int64_t at = (int64_t)f[i] - 1;
      return fail(h, NERFHIP_E_INVALID, "dtype f16x3: a weight of network " + std::to_string(i) + " (" + param_name(h->nets[i].L, at) +
                                            ", flat index " + std::to_string(h->nets[i].param_off + at) +
                                            ") is non-finite or reaches |w| >= 1023.5, beyond the split-f16 operand range (use dtype f32)");
    }
  return fail(h, NERFHIP_E_INVALID, "dtype f16x3: an activation left the f16 operand range (|x| > 65504) and the network produced non-finite "
                                    "outputs; the results of the calls since the last check are invalid (use dtype f32 for these weights)");
}

int check_handle(nerfhip* h) {
  if (!h) return fail(nullptr, NERFHIP_E_INVALID, "null handle");
  hipError_t e = hipSetDevice(h->cfg.device);
  if (e != hipSuccess) return fail(h, NERFHIP_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return NERFHIP_OK;
}

}  // namespace

// training path lives in train.hip (same translation-unit-private structs are shared through this header-less
// include to keep the handle layout in one place)
#include "train_impl.inc"
#include "train_bf16.inc"
#include "dp_rccl.inc"

extern "C" {

const char* nerfhip_last_error(const nerfhip_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int nerfhip_create(const nerfhip_config* cfg, nerfhip_t** out) {
  if (!cfg || !out) return fail(nullptr, NERFHIP_E_INVALID, "null argument");
  *out = nullptr;
  const nerfhip_config& c = *cfg;
  auto bad = [&](const char* m) { return fail(nullptr, NERFHIP_E_INVALID, m); };
  if (!(c.W == 32 || c.W == 64 || c.W == 128 || c.W == 256)) return bad("W must be 32, 64, 128 or 256");
  if (c.D < 1 || c.D > kMaxD) return bad("D must be in [1,16]");
  if (c.pos_l < 0 || c.pos_l > 10) return bad("posL must be in [0,10]");
  if (c.use_dir && !c.use_sh && (c.dir_l < 0 || c.dir_l > 4)) return bad("dirL must be in [0,4]");
  if (c.n_skips < 0 || c.n_skips > NERFHIP_MAX_SKIPS) return bad("too many skips");
  for (int k = 0; k < c.n_skips; ++k)
    if (c.skips[k] < 0 || c.skips[k] > c.D - 2) return bad("each skip must be in [0, D-2]");
  if (c.n_samples < 3 || c.n_samples > 1024) return bad("NSamples must be in [3,1024]");
  if (c.use_hierarchical && (c.n_importance < 1 || c.n_samples + c.n_importance > 2048))
    return bad("NImportance must be >= 1 and NSamples+NImportance <= 2048");
  if (c.max_rays < 1) return bad("max_rays must be >= 1");
  if (c.dtype != NERFHIP_F32 && c.dtype != NERFHIP_BF16 && c.dtype != NERFHIP_F16X3) return bad("unknown dtype");
  if (c.dtype == NERFHIP_BF16 || c.dtype == NERFHIP_F16X3) {   // limits of the 16-bit-operand kernels (tile indices are 32-bit; W/32 in {2,4,8})
    if (c.W == 32) return bad("dtype bf16 / f16x3 needs W = 64, 128 or 256");
    const long long smax = c.use_hierarchical ? c.n_samples + c.n_importance : c.n_samples;
    if ((long long)c.max_rays * smax >= (1LL << 31)) return bad("dtype bf16 / f16x3: max_rays x samples per ray must be < 2^31");
  }
  if (c.lrate_decay <= 0) return bad("lrateDecay must be > 0");
  { // per-ray kernels stage one ray per wave in at most 64 KiB of LDS per 4-wave workgroup (ray_kernels.hip)
    const int S_all = c.use_hierarchical ? c.n_samples + c.n_importance : c.n_samples;
    if (composite_bwd_lds_bytes(S_all) > 64 * 1024)
      return bad("NSamples + NImportance too large for the compositing backward kernel (48 B of LDS per sample, 64 KiB)");
    if (c.use_hierarchical && sample_pdf_lds_bytes(c.n_samples, c.n_importance, c.perturb) > 64 * 1024)
      return bad("NSamples / NImportance too large for the resampling kernel (CDF, bins and sort buffer must fit 64 KiB of LDS)");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(nullptr, NERFHIP_E_HIP, "no HIP device available (libnerfhip has no CPU path)");
  if (c.device < 0 || c.device >= ndev) return bad("device ordinal out of range");
  nerfhip* h = new nerfhip();
  h->cfg = c;
  if (hipSetDevice(c.device) != hipSuccess) { delete h; return fail(nullptr, NERFHIP_E_HIP, "hipSetDevice failed"); }
  { int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c.device) == hipSuccess && ncu > 0) h->n_cu = ncu; }
  h->nnets = c.use_hierarchical ? 2 : 1;
  h->N = c.n_samples;
  h->NI = c.use_hierarchical ? c.n_importance : 0;
  h->S = h->N + h->NI;
  h->R = c.max_rays;
  const NetLayout L = make_layout(c);
  h->nparams = L.count * h->nnets;
  auto cleanup_fail = [&](int code, const std::string& m) { g_create_error = m; nerfhip_destroy(h); return code; };
#define CCHK(expr)                                                                                        \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess)                                                      \
      return cleanup_fail(e_ == hipErrorOutOfMemory ? NERFHIP_E_NOMEM : NERFHIP_E_HIP,                    \
                          std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
  // kernels with > 64 KiB of dynamic LDS: raise the limit on THIS device (per handle, not per process)
  CCHK(prepare_wgrad_f32());
  CCHK(prepare_wgrad_f16x3());
  if (c.dtype == NERFHIP_BF16) {
    CCHK(prepare_wgrad_bf16());
    CCHK(prepare_mlp_fwd_bf16(c.W, L.mode));
    CCHK(prepare_mlp_fwd_bf16_t16(c.W, L.mode));
    CCHK(prepare_mlp_bwd_bf16(c.W, L.mode));
  }
  if (c.dtype == NERFHIP_F16X3) { CCHK(prepare_mlp_fwd_f16x3(c.W, L.mode)); CCHK(prepare_mlp_fwd_f16x3_t16(c.W, L.mode)); CCHK(prepare_mlp_bwd_f16x3(c.W, L.mode)); }
  CCHK(hipGetSymbolAddress((void**)&h->static_scratch, HIP_SYMBOL(scratch)));   // this device's copy of the array
  CCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  CCHK(dalloc(&h->d_params, h->nparams));
  CCHK(dalloc(&h->d_grads, h->nparams));
  CCHK(dalloc(&h->d_m, h->nparams));
  CCHK(dalloc(&h->d_v, h->nparams));
  CCHK(hipMemsetAsync(h->d_params, 0, h->nparams * sizeof(float), h->stream));
  CCHK(hipMemsetAsync(h->d_grads, 0, h->nparams * sizeof(float), h->stream));
  CCHK(hipMemsetAsync(h->d_m, 0, h->nparams * sizeof(float), h->stream));
  CCHK(hipMemsetAsync(h->d_v, 0, h->nparams * sizeof(float), h->stream));
  CCHK(dalloc(&h->d_loss, 3));
  for (int i = 0; i < h->nnets; ++i) {
    Net& n = h->nets[i];
    n.L = L;
    n.WT = c.W / 32;
    n.HT = (L.head_rows + 31) / 32;
    n.param_off = (int64_t)i * L.count;
    build_chunks(n);
    CCHK(dalloc(&n.d_fwd_desc, n.fwd_desc.size()));
    CCHK(dalloc(&n.d_bwd_desc, std::max<size_t>(n.bwd_desc.size(), 1)));
    CCHK(dalloc(&n.d_off_fwd, n.off_fwd.size()));
    CCHK(dalloc(&n.d_off_bwd, n.off_bwd.size()));
    CCHK(hipMemcpy(n.d_fwd_desc, n.fwd_desc.data(), n.fwd_desc.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
    if (!n.bwd_desc.empty())
      CCHK(hipMemcpy(n.d_bwd_desc, n.bwd_desc.data(), n.bwd_desc.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice));
    CCHK(hipMemcpy(n.d_off_fwd, n.off_fwd.data(), n.off_fwd.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    CCHK(hipMemcpy(n.d_off_bwd, n.off_bwd.data(), n.off_bwd.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    // one chunk of padding behind each stream: the kernels' fragment prefetch runs a few tiles past the end
    const size_t pad = (size_t)32 * c.W;
    CCHK(dalloc(&n.d_wpack_fwd, (size_t)n.off_fwd.back() + pad));
    CCHK(dalloc(&n.d_wpack_bwd, (size_t)n.off_bwd.back() + pad));
    CCHK(hipMemsetAsync(n.d_wpack_fwd, 0, ((size_t)n.off_fwd.back() + pad) * sizeof(float), h->stream));
    CCHK(hipMemsetAsync(n.d_wpack_bwd, 0, ((size_t)n.off_bwd.back() + pad) * sizeof(float), h->stream));
    CCHK(dalloc(&n.d_vecpack, (size_t)vecpack_floats(L.D, L.W)));
    if (c.dtype == NERFHIP_BF16) {
      build_bf16_streams(n);
      const size_t padf = 4 * 32;   // the LDS ring requests up to 3 blocks past the end
      CCHK(dalloc(&n.d_bf_full_desc, n.bf_full.size()));
      CCHK(dalloc(&n.d_bf_sigma_desc, n.bf_sigma.size()));
      CCHK(hipMemcpy(n.d_bf_full_desc, n.bf_full.data(), n.bf_full.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMemcpy(n.d_bf_sigma_desc, n.bf_sigma.data(), n.bf_sigma.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMalloc(&n.d_bf_full, (n.bf_full.size() + padf) * 1024));
      CCHK(hipMalloc(&n.d_bf_sigma, (n.bf_sigma.size() + padf) * 1024));
      CCHK(hipMemsetAsync(n.d_bf_full, 0, (n.bf_full.size() + padf) * 1024, h->stream));
      CCHK(hipMemsetAsync(n.d_bf_sigma, 0, (n.bf_sigma.size() + padf) * 1024, h->stream));
      CCHK(dalloc(&n.d_bf_full16_desc, n.bf_full16.size()));
      CCHK(dalloc(&n.d_bf_sigma16_desc, n.bf_sigma16.size()));
      CCHK(hipMemcpy(n.d_bf_full16_desc, n.bf_full16.data(), n.bf_full16.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMemcpy(n.d_bf_sigma16_desc, n.bf_sigma16.data(), n.bf_sigma16.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMalloc(&n.d_bf_full16, (n.bf_full16.size() + padf) * 1024));
      CCHK(hipMalloc(&n.d_bf_sigma16, (n.bf_sigma16.size() + padf) * 1024));
      CCHK(hipMemsetAsync(n.d_bf_full16, 0, (n.bf_full16.size() + padf) * 1024, h->stream));
      CCHK(hipMemsetAsync(n.d_bf_sigma16, 0, (n.bf_sigma16.size() + padf) * 1024, h->stream));
      CCHK(dalloc(&n.d_bvec, (size_t)L.D * L.W + 5 * 32 + 32));
      CCHK(dalloc(&n.d_bf_bwd_desc, n.bf_bwd.size()));
      CCHK(hipMemcpy(n.d_bf_bwd_desc, n.bf_bwd.data(), n.bf_bwd.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMalloc(&n.d_bf_bwd, (n.bf_bwd.size() + padf) * 1024));
      CCHK(hipMemsetAsync(n.d_bf_bwd, 0, (n.bf_bwd.size() + padf) * 1024, h->stream));
    }
    if (c.dtype == NERFHIP_F16X3) {
      build_fwd_streams(n, 16, n.x3_full, n.x3_sigma);
      build_bwd_stream(n, 16, n.x3_bwd);
      const size_t padf = 4 * 32;   // the LDS ring requests up to 3 blocks (of 32 fragments) past the end
      CCHK(dalloc(&n.d_x3_bwd_desc, n.x3_bwd.size()));
      CCHK(hipMemcpy(n.d_x3_bwd_desc, n.x3_bwd.data(), n.x3_bwd.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMalloc(&n.d_x3_bwd, (2 * n.x3_bwd.size() + padf) * 1024));
      CCHK(hipMemsetAsync(n.d_x3_bwd, 0, (2 * n.x3_bwd.size() + padf) * 1024, h->stream));
      CCHK(dalloc(&n.d_x3_full_desc, n.x3_full.size()));
      CCHK(dalloc(&n.d_x3_sigma_desc, n.x3_sigma.size()));
      CCHK(hipMemcpy(n.d_x3_full_desc, n.x3_full.data(), n.x3_full.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMemcpy(n.d_x3_sigma_desc, n.x3_sigma.data(), n.x3_sigma.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMalloc(&n.d_x3_full, (2 * n.x3_full.size() + padf) * 1024));
      CCHK(hipMalloc(&n.d_x3_sigma, (2 * n.x3_sigma.size() + padf) * 1024));
      CCHK(hipMemsetAsync(n.d_x3_full, 0, (2 * n.x3_full.size() + padf) * 1024, h->stream));
      CCHK(hipMemsetAsync(n.d_x3_sigma, 0, (2 * n.x3_sigma.size() + padf) * 1024, h->stream));
      build_fwd_streams_t16(n, 16, n.x3_full16, n.x3_sigma16);
      CCHK(dalloc(&n.d_x3_full16_desc, n.x3_full16.size()));
      CCHK(dalloc(&n.d_x3_sigma16_desc, n.x3_sigma16.size()));
      CCHK(hipMemcpy(n.d_x3_full16_desc, n.x3_full16.data(), n.x3_full16.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMemcpy(n.d_x3_sigma16_desc, n.x3_sigma16.data(), n.x3_sigma16.size() * sizeof(FragDesc), hipMemcpyHostToDevice));
      CCHK(hipMalloc(&n.d_x3_full16, (2 * n.x3_full16.size() + padf) * 1024));
      CCHK(hipMalloc(&n.d_x3_sigma16, (2 * n.x3_sigma16.size() + padf) * 1024));
      CCHK(hipMemsetAsync(n.d_x3_full16, 0, (2 * n.x3_full16.size() + padf) * 1024, h->stream));
      CCHK(hipMemsetAsync(n.d_x3_sigma16, 0, (2 * n.x3_sigma16.size() + padf) * 1024, h->stream));
      CCHK(dalloc(&n.d_bvec, (size_t)L.D * L.W + 5 * 32 + 32));
      if (!h->d_x3flags) {
        CCHK(dalloc(&h->d_x3flags, 4));
        CCHK(hipMemsetAsync(h->d_x3flags, 0, 4 * sizeof(uint32_t), h->stream));
      }
    }
    CCHK(dalloc(&n.d_dirfeat, (size_t)h->R * std::max(32 * n.HT, 12)));
  }
  const size_t R = (size_t)h->R, N = h->N, S = h->S, NI = std::max(h->NI, 1);
  CCHK(dalloc(&h->w_o, R * 3)); CCHK(dalloc(&h->w_d, R * 3)); CCHK(dalloc(&h->w_b, R * 2)); CCHK(dalloc(&h->w_v, R * 3));
  CCHK(dalloc(&h->w_label, R * 3)); CCHK(dalloc(&h->w_rgb_f, R * 3)); CCHK(dalloc(&h->w_rgb_c, R * 3));
  CCHK(dalloc(&h->w_depth, R)); CCHK(dalloc(&h->w_acc, R)); CCHK(dalloc(&h->w_drgb, R * 3));
  CCHK(dalloc(&h->w_zc, R * N)); CCHK(dalloc(&h->w_sigc, R * N)); CCHK(dalloc(&h->w_wc, R * N));
  CCHK(dalloc(&h->w_crawc, R * N * 3));
  CCHK(dalloc(&h->w_zf, R * S)); CCHK(dalloc(&h->w_sigf, R * S)); CCHK(dalloc(&h->w_wf, R * S));
  CCHK(dalloc(&h->w_crawf, R * S * 3));
  CCHK(dalloc(&h->w_trand, R * N)); CCHK(dalloc(&h->w_urand, R * NI)); CCHK(dalloc(&h->w_noisec, R * N));
  CCHK(dalloc(&h->w_noisef, R * S));
  CCHK(dalloc(&h->w_u8, R * 3));
  CCHK(hipStreamSynchronize(h->stream));
#undef CCHK
  *out = h;
  return NERFHIP_OK;
}

void nerfhip_destroy(nerfhip_t* h) {
  if (!h) return;
  hipSetDevice(h->cfg.device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->dp_comm) nerfhip_dp_destroy(h);
  auto fr = [](void* p) { if (p) hipFree(p); };
  fr(h->d_params); fr(h->d_grads); fr(h->d_m); fr(h->d_v); fr(h->d_loss); fr(h->d_items); fr(h->d_items_first); fr(h->d_bitems); fr(h->d_amax); fr(h->d_x3flags);
  for (int i = 0; i < 2; ++i) {
    Net& n = h->nets[i];
    fr(n.d_fwd_desc); fr(n.d_bwd_desc); fr(n.d_off_fwd); fr(n.d_off_bwd); fr(n.d_wpack_fwd); fr(n.d_wpack_bwd); fr(n.d_bf_full_desc); fr(n.d_bf_sigma_desc); fr(n.d_bf_full); fr(n.d_bf_sigma); fr(n.d_bf_full16_desc); fr(n.d_bf_sigma16_desc); fr(n.d_bf_full16); fr(n.d_bf_sigma16); fr(n.d_bvec); fr(n.d_bf_bwd_desc); fr(n.d_bf_bwd); fr(n.d_x3_full_desc); fr(n.d_x3_sigma_desc); fr(n.d_x3_bwd_desc); fr(n.d_x3_full); fr(n.d_x3_sigma); fr(n.d_x3_full16_desc); fr(n.d_x3_sigma16_desc); fr(n.d_x3_full16); fr(n.d_x3_sigma16); fr(n.d_x3_bwd); fr(n.bst.enc); fr(n.bst.act); fr(n.bst.hc); fr(n.bst.dy); fr(n.bst.dh); fr(n.bst.encdir); fr(n.bst.relu);
    fr(n.d_vecpack); fr(n.d_dirfeat); fr(n.d_encdirT); fr(n.d_ddirT);
    fr(n.st.enc); fr(n.st.act); fr(n.st.hc); fr(n.st.dy); fr(n.st.dh); fr(n.st.mask); fr(n.d_dsigma); fr(n.d_dcraw);
  }
  fr(h->w_o); fr(h->w_d); fr(h->w_b); fr(h->w_v); fr(h->w_label); fr(h->w_rgb_f); fr(h->w_rgb_c); fr(h->w_depth);
  fr(h->w_acc); fr(h->w_drgb); fr(h->w_zc); fr(h->w_sigc); fr(h->w_wc); fr(h->w_crawc); fr(h->w_zf); fr(h->w_sigf);
  fr(h->w_wf); fr(h->w_crawf); fr(h->w_trand); fr(h->w_urand); fr(h->w_noisec); fr(h->w_noisef); fr(h->w_u8);
  for (auto& r : h->prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  for (auto e : h->ev_pool) hipEventDestroy(e);
  if (h->d_pose_rays) hipFree(h->d_pose_rays);
  if (h->own_stream) hipStreamDestroy(h->own_stream);
  delete h;
}

int nerfhip_set_stream(nerfhip_t* h, void* s) {
  int rc = check_handle(h); if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->stream = s ? (hipStream_t)s : h->own_stream;
  return NERFHIP_OK;
}

int nerfhip_synchronize(nerfhip_t* h) {
  int rc = check_handle(h); if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_x3_range(h);   // what the asynchronous calls since the last check flagged (dtype f16x3)
}

int64_t nerfhip_param_count(const nerfhip_t* h) { return h ? h->nparams : 0; }

int nerfhip_set_params(nerfhip_t* h, const float* src, int64_t n) {
  int rc = check_handle(h); if (rc) return rc;
  if (!src || n != h->nparams) return fail(h, NERFHIP_E_INVALID, "set_params: wrong size");
  HIPCHK(h, hipMemcpyAsync(h->d_params, src, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  rc = repack(h);
  if (rc) return rc;
  return check_x3_range(h);   // dtype f16x3: weights beyond the split-f16 range are rejected here, by name
}

int nerfhip_get_params(nerfhip_t* h, float* dst, int64_t n) {
  int rc = check_handle(h); if (rc) return rc;
  if (!dst || n != h->nparams) return fail(h, NERFHIP_E_INVALID, "get_params: wrong size");
  HIPCHK(h, hipMemcpyAsync(dst, h->d_params, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_x3_range(h);
}

int nerfhip_init_params(nerfhip_t* h, uint64_t seed) {
  int rc = check_handle(h); if (rc) return rc;
  for (int i = 0; i < h->nnets; ++i) {
    const Net& n = h->nets[i];
    const NetLayout& L = n.L;
    auto seg = [&](int64_t off, int64_t cnt, int fan_in) {
      return launch_init(h->d_params, n.param_off + off, cnt, fan_in > 0 ? (float)std::sqrt(2.0 / fan_in) : 0.f, seed,
                         h->stream);
    };
    for (int l = 0; l < L.D; ++l) {
      HIPCHK(h, seg(L.w_off[l], (int64_t)L.W * L.fan_in[l], L.fan_in[l]));
      HIPCHK(h, seg(L.b_off[l], L.W, 0));
    }
    HIPCHK(h, seg(L.ws_off, L.W, L.W));
    HIPCHK(h, seg(L.bs_off, 1, 0));
    HIPCHK(h, seg(L.wf_off, (int64_t)L.head_rows * L.W, L.W));
    HIPCHK(h, seg(L.bf_off, L.head_rows, 0));
    if (L.mode == 1) {
      HIPCHK(h, seg(L.wd_off, (int64_t)L.H2 * L.Ed, L.Ed));
      HIPCHK(h, seg(L.bd_off, L.H2, 0));
      HIPCHK(h, seg(L.wc_off, 3 * L.H2, L.H2));
      HIPCHK(h, seg(L.bc_off, 3, 0));
    }
  }
  HIPCHK(h, hipMemsetAsync(h->d_m, 0, h->nparams * sizeof(float), h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_v, 0, h->nparams * sizeof(float), h->stream));
  h->opt_step = 0;
  return repack(h);
}

int nerfhip_get_opt_state(nerfhip_t* h, float* m, float* v, int64_t n, int64_t* step) {
  int rc = check_handle(h); if (rc) return rc;
  if (n != h->nparams) return fail(h, NERFHIP_E_INVALID, "get_opt_state: wrong size");
  if (m) HIPCHK(h, hipMemcpyAsync(m, h->d_m, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (v) HIPCHK(h, hipMemcpyAsync(v, h->d_v, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (step) *step = h->opt_step;
  return NERFHIP_OK;
}

int nerfhip_set_opt_state(nerfhip_t* h, const float* m, const float* v, int64_t n, int64_t step) {
  int rc = check_handle(h); if (rc) return rc;
  if (n != h->nparams || step < 0) return fail(h, NERFHIP_E_INVALID, "set_opt_state: bad argument");
  if (m) HIPCHK(h, hipMemcpyAsync(h->d_m, m, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if (v) HIPCHK(h, hipMemcpyAsync(h->d_v, v, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->opt_step = step;
  return NERFHIP_OK;
}

// ---- render -------------------------------------------------------------------------------------
int nerfhip_render_dev(nerfhip_t* h, const float* o, const float* d, const float* b, const float* v, int64_t n,
                       float* rgb, nerfhip_aux* aux) {
  int rc = check_handle(h); if (rc) return rc;
  if (!o || !d || !b || !v || !rgb) return fail(h, NERFHIP_E_INVALID, "render: null pointer");
  DevIn in{o, d, b, v, aux ? aux->t_rand : nullptr, aux ? aux->u_rand : nullptr, nullptr, nullptr};
  rc = forward_dev(h, in, n, false, false);
  if (rc) return rc;
  rc = copy_out(h, rgb, h->w_rgb_f, n * 3 * sizeof(float), false);
  if (rc) return rc;
  return export_aux(h, aux, n, false, false);
}

// bytes of static scratch one ray of a render needs (every piece rounded up to 256 B per call, see below)
static size_t static_floats_per_ray(const nerfhip* h) {
  const size_t N = h->N, S = h->S;
  size_t f = 3 + 1 + 1;                                       // rgb, depth, acc
  f += S * (1 + 1 + 1 + 3);                                   // z, sigma, w, craw of the fine pass
  if (h->cfg.use_hierarchical) f += N * 3;                    // z, sigma, w of the coarse pass
  f += (size_t)std::max(32 * h->nets[h->nnets - 1].HT, 12);   // direction features of the fine network
  return f;
}

int64_t nerfhip_render_static_max_rays(const nerfhip_t* h) {
  if (!h) return 0;
  return (int64_t)((NERFHIP_STATIC_SCRATCH_BYTES - 16 * 256) / (static_floats_per_ray(h) * sizeof(float)));
}

int nerfhip_render_dev_static(nerfhip_t* h, const float* o, const float* d, const float* b, const float* v, int64_t n,
                              float* rgb) {
  int rc = check_handle(h); if (rc) return rc;
  if (!o || !d || !b || !v || !rgb) return fail(h, NERFHIP_E_INVALID, "render_dev_static: null pointer");
  if (n <= 0 || n > nerfhip_render_static_max_rays(h))
    return fail(h, NERFHIP_E_INVALID, "render_dev_static: n exceeds the static scratch (nerfhip_render_static_max_rays: " +
                                          std::to_string(nerfhip_render_static_max_rays(h)) + " rays for this configuration)");
  if (h->cfg.dtype == NERFHIP_BF16 && n * (int64_t)h->S >= (int64_t)1 << 31)
    return fail(h, NERFHIP_E_INVALID, "render_dev_static: rays x samples must be < 2^31 (bf16 kernels)");
  if (!h->static_scratch) return fail(h, NERFHIP_E_STATE, "render_dev_static: no scratch address for this device");
  // carve the per-call intermediates out of THIS device's scratch array (pointer from hipGetSymbolAddress); nothing
  // in it has to start at zero: every element the kernels read is written earlier in the same call
  unsigned char* p = h->static_scratch;
  auto take = [&](size_t floats) { float* r = (float*)p; p += (floats * sizeof(float) + 255) / 256 * 256; return r; };
  const size_t R = (size_t)n, N = h->N, S = h->S;
  Net& fine = h->nets[h->nnets - 1];
  struct Saved { float *rgb_f, *depth, *acc, *zc, *sigc, *wc, *zf, *sigf, *wf, *crawf, *dirfeat; bool prof; } sv{
      h->w_rgb_f, h->w_depth, h->w_acc, h->w_zc, h->w_sigc, h->w_wc, h->w_zf, h->w_sigf, h->w_wf, h->w_crawf, fine.d_dirfeat,
      h->prof_on};
  h->w_rgb_f = take(R * 3); h->w_depth = take(R); h->w_acc = take(R);
  if (h->cfg.use_hierarchical) {
    h->w_zc = take(R * N); h->w_sigc = take(R * N); h->w_wc = take(R * N);
    h->w_zf = take(R * S);
  } else {
    h->w_zc = take(R * S);   // without a fine pass the coarse depths ARE the fine depths (forward_dev)
  }
  h->w_sigf = take(R * S); h->w_wf = take(R * S); h->w_crawf = take(R * S * 3);
  fine.d_dirfeat = take(R * (size_t)std::max(32 * fine.HT, 12));
  h->prof_on = false;        // profiling scopes create events: this entry point allocates nothing
  if ((size_t)(p - h->static_scratch) > NERFHIP_STATIC_SCRATCH_BYTES) rc = fail(h, NERFHIP_E_INVALID, "render_dev_static: scratch overflow");
  if (!rc) {
    DevIn in{o, d, b, v, nullptr, nullptr, nullptr, nullptr};
    rc = forward_dev(h, in, n, false, false, /*own_bounds=*/true);
  }
  if (!rc) rc = copy_out(h, rgb, h->w_rgb_f, (size_t)n * 3 * sizeof(float), false);
  h->w_rgb_f = sv.rgb_f; h->w_depth = sv.depth; h->w_acc = sv.acc; h->w_zc = sv.zc; h->w_sigc = sv.sigc; h->w_wc = sv.wc;
  h->w_zf = sv.zf; h->w_sigf = sv.sigf; h->w_wf = sv.wf; h->w_crawf = sv.crawf; fine.d_dirfeat = sv.dirfeat;
  h->prof_on = sv.prof;
  return rc;
}

int nerfhip_render(nerfhip_t* h, const float* o, const float* d, const float* b, const float* v, int64_t n, float* rgb,
                   nerfhip_aux* aux) {
  int rc = check_handle(h); if (rc) return rc;
  if (!o || !d || !b || !v || !rgb || n < 0) return fail(h, NERFHIP_E_INVALID, "render: bad argument");
  for (int64_t off = 0; off < n; off += h->R) {
    const int64_t nr = std::min<int64_t>(h->R, n - off);
    DevIn in;
    rc = stage_in(h, o, d, b, v, nullptr, aux, off, nr, &in);
    if (rc) return rc;
    rc = forward_dev(h, in, nr, false, false);
    if (rc) return rc;
    rc = copy_out(h, rgb + off * 3, h->w_rgb_f, nr * 3 * sizeof(float), true);
    if (rc) return rc;
    nerfhip_aux s = aux_slice(aux, off, h->N, h->S);
    rc = export_aux(h, aux ? &s : nullptr, nr, false, true);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return check_x3_range(h);
}

int nerfhip_render_u8(nerfhip_t* h, const float* o, const float* d, const float* b, const float* v, int64_t n,
                      uint8_t* out) {
  int rc = check_handle(h); if (rc) return rc;
  if (!o || !d || !b || !v || !out || n < 0) return fail(h, NERFHIP_E_INVALID, "render_u8: bad argument");
  for (int64_t off = 0; off < n; off += h->R) {
    const int64_t nr = std::min<int64_t>(h->R, n - off);
    DevIn in;
    rc = stage_in(h, o, d, b, v, nullptr, nullptr, off, nr, &in);
    if (rc) return rc;
    rc = forward_dev(h, in, nr, false, false);
    if (rc) return rc;
    HIPCHK(h, launch_to_u8(h->w_rgb_f, nr * 3, h->w_u8, h->stream));
    HIPCHK(h, hipMemcpyAsync(out + off * 3, h->w_u8, nr * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return check_x3_range(h);
}

// rows [row0, row1) of the image: one rank's shard of a frame whose rows are spread over several GPUs (BASELINE
// configs[2]); the whole frame is rows [0, H)
int nerfhip_render_pose_rows_u8(nerfhip_t* h, int32_t H, int32_t W, float focal, const float* c2w, float near, float far,
                                int32_t ndc, int32_t row0, int32_t row1, uint8_t* out) {
  int rc = check_handle(h); if (rc) return rc;
  if (!c2w || H <= 0 || W <= 0 || !(focal > 0.f)) return fail(h, NERFHIP_E_INVALID, "render_pose_u8: bad argument");
  if (row0 < 0 || row1 < row0 || row1 > H) return fail(h, NERFHIP_E_INVALID, "render_pose_rows_u8: rows must satisfy 0 <= row0 <= row1 <= H");
  const int64_t n = (int64_t)(row1 - row0) * W;
  if (n == 0) return NERFHIP_OK;       // a rank without rows (more ranks than rows) renders nothing
  if (!out) return fail(h, NERFHIP_E_INVALID, "render_pose_u8: null output");
  if (h->pose_rays_cap < n) {   // block-sized ray buffers, kept for the next call of this size
    if (h->d_pose_rays) { HIPCHK(h, hipFree(h->d_pose_rays)); h->d_pose_rays = nullptr; h->pose_rays_cap = 0; }
    HIPCHK(h, hipMalloc((void**)&h->d_pose_rays, (size_t)n * 8 * sizeof(float)));
    h->pose_rays_cap = n;
  }
  float* ro = h->d_pose_rays;
  float* rd = ro + n * 3;
  float* bd = rd + n * 3;
  HIPCHK(h, launch_gen_rays(H, W, focal, c2w, near, far, ndc, row0, row1 - row0, ro, rd, bd, h->stream));
  for (int64_t off = 0; off < n; off += h->R) {
    const int64_t nr = std::min<int64_t>(h->R, n - off);
    DevIn in{ro + off * 3, rd + off * 3, bd + off * 2, rd + off * 3, nullptr, nullptr, nullptr, nullptr};
    rc = forward_dev(h, in, nr, false, false);
    if (rc) return rc;
    HIPCHK(h, launch_to_u8(h->w_rgb_f, nr * 3, h->w_u8, h->stream));
    HIPCHK(h, hipMemcpyAsync(out + off * 3, h->w_u8, nr * 3, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));   // w_u8 is reused by the next block
  }
  return check_x3_range(h);
}

int nerfhip_render_pose_u8(nerfhip_t* h, int32_t H, int32_t W, float focal, const float* c2w, float near, float far,
                           int32_t ndc, uint8_t* out) {
  if (h && !out) return fail(h, NERFHIP_E_INVALID, "render_pose_u8: bad argument");
  return nerfhip_render_pose_rows_u8(h, H, W, focal, c2w, near, far, ndc, 0, H, out);
}

int nerfhip_forward_train(nerfhip_t* h, const float* o, const float* d, const float* b, const float* v, int64_t n,
                          float* rgb_fine, float* rgb_coarse, nerfhip_aux* aux) {
  int rc = check_handle(h); if (rc) return rc;
  if (!o || !d || !b || !v || !rgb_fine || n < 0) return fail(h, NERFHIP_E_INVALID, "forward_train: bad argument");
  for (int64_t off = 0; off < n; off += h->R) {
    const int64_t nr = std::min<int64_t>(h->R, n - off);
    DevIn in;
    rc = stage_in(h, o, d, b, v, nullptr, aux, off, nr, &in);
    if (rc) return rc;
    rc = forward_dev(h, in, nr, true, false);
    if (rc) return rc;
    rc = copy_out(h, rgb_fine + off * 3, h->w_rgb_f, nr * 3 * sizeof(float), true);
    if (rc) return rc;
    if (rgb_coarse && h->cfg.use_hierarchical) {
      rc = copy_out(h, rgb_coarse + off * 3, h->w_rgb_c, nr * 3 * sizeof(float), true);
      if (rc) return rc;
    }
    nerfhip_aux s = aux_slice(aux, off, h->N, h->S);
    rc = export_aux(h, aux ? &s : nullptr, nr, true, true);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return check_x3_range(h);
}

// ---- single kernels -------------------------------------------------------------------------------
int nerfhip_k_stratified(nerfhip_t* h, const float* bounds, const float* t_rand, int64_t n, float* z) {
  int rc = check_handle(h); if (rc) return rc;
  if (!bounds || !z || n <= 0) return fail(h, NERFHIP_E_INVALID, "k_stratified: bad argument");
  HIPCHK(h, launch_stratified(bounds, t_rand, n, h->N, h->cfg.lin_disp, h->cfg.perturb, h->cfg.seed, h->rng_step++, z,
                              h->stream));
  return NERFHIP_OK;
}

int nerfhip_k_composite_fwd(nerfhip_t* h, const float* sigma, const float* noise, const float* craw, const float* z,
                            int64_t n, int32_t S, float* w, float* rgb, float* depth, float* acc) {
  int rc = check_handle(h); if (rc) return rc;
  if (!sigma || !z || n <= 0 || S < 2) return fail(h, NERFHIP_E_INVALID, "k_composite_fwd: bad argument");
  HIPCHK(h, launch_composite_fwd(sigma, noise, craw, z, n, S, noise ? h->cfg.raw_noise_std : 0.f, h->cfg.white_bkgd,
                                 h->cfg.seed, 0, kStreamNoiseF, w, rgb, depth, acc, h->stream));
  return NERFHIP_OK;
}

int nerfhip_k_composite_bwd(nerfhip_t* h, const float* sigma, const float* noise, const float* craw, const float* z,
                            const float* d_rgb, int64_t n, int32_t S, float* d_sigma, float* d_craw) {
  int rc = check_handle(h); if (rc) return rc;
  if (!sigma || !z || !craw || !d_rgb || !d_sigma || !d_craw || n <= 0 || S < 2)
    return fail(h, NERFHIP_E_INVALID, "k_composite_bwd: bad argument");
  HIPCHK(h, launch_composite_bwd(sigma, noise, craw, z, d_rgb, n, S, noise ? h->cfg.raw_noise_std : 0.f,
                                 h->cfg.white_bkgd, h->cfg.seed, 0, kStreamNoiseF, d_sigma, d_craw, h->stream));
  return NERFHIP_OK;
}

int nerfhip_k_sample_pdf(nerfhip_t* h, const float* w, const float* z, const float* u_rand, int64_t n, float* cdf,
                         int32_t* ind, float* z_samples, float* z_fine) {
  int rc = check_handle(h); if (rc) return rc;
  if (!w || !z || !z_fine || n <= 0 || h->NI < 1) return fail(h, NERFHIP_E_INVALID, "k_sample_pdf: bad argument");
  HIPCHK(h, launch_sample_pdf(w, z, u_rand, n, h->N, h->NI, h->cfg.perturb, h->cfg.seed, h->rng_step++, cdf, ind,
                              z_samples, z_fine, h->stream));
  return NERFHIP_OK;
}

int nerfhip_k_mlp_forward(nerfhip_t* h, int32_t net, const float* o, const float* d, const float* v, const float* z,
                          int64_t n, int32_t S, float* sigma, float* craw) {
  int rc = check_handle(h); if (rc) return rc;
  if (net < 0 || net >= h->nnets || !o || !d || !v || !z || !sigma || n <= 0 || n > h->R || S < 1)
    return fail(h, NERFHIP_E_INVALID, "k_mlp_forward: bad argument");
  if (!h->packed) { rc = repack(h); if (rc) return rc; }
  Net& nn = h->nets[net];
  if (craw) { rc = dir_features(h, nn, v, n, false); if (rc) return rc; }
  return mlp_forward(h, nn, false, o, d, z, n, S, sigma, craw, craw != nullptr);
}

int nerfhip_k_gen_rays(nerfhip_t* h, int32_t H, int32_t W, float focal, const float* c2w, float near, float far,
                       int32_t ndc, float* rays_o, float* rays_d, float* bounds) {
  int rc = check_handle(h); if (rc) return rc;
  if (H < 1 || W < 1 || !c2w || !rays_o || !rays_d || !bounds) return fail(h, NERFHIP_E_INVALID, "k_gen_rays: bad argument");
  HIPCHK(h, launch_gen_rays(H, W, focal, c2w, near, far, ndc, 0, H, rays_o, rays_d, bounds, h->stream));
  return NERFHIP_OK;
}

int nerfhip_k_random(nerfhip_t* h, int32_t stream, uint64_t seed, uint32_t step, int64_t n, float* out) {
  int rc = check_handle(h); if (rc) return rc;
  if (!out || n <= 0 || stream < NERFHIP_RNG_PERTURB || stream > NERFHIP_RNG_INIT)
    return fail(h, NERFHIP_E_INVALID, "k_random: bad argument");
  const int normal = stream == NERFHIP_RNG_NOISE_COARSE || stream == NERFHIP_RNG_NOISE_FINE || stream == NERFHIP_RNG_INIT;
  HIPCHK(h, launch_random(seed, (uint32_t)stream, step, n, normal, out, h->stream));
  return NERFHIP_OK;
}

int nerfhip_rng_step(const nerfhip_t* h, uint32_t* last, uint32_t* next) {
  if (!h) return NERFHIP_E_INVALID;
  if (last) *last = h->last_rng_step;
  if (next) *next = h->rng_step;
  return NERFHIP_OK;
}

// ---- profiling ------------------------------------------------------------------------------------
int nerfhip_set_profiling(nerfhip_t* h, int32_t on) {
  if (!h) return NERFHIP_E_INVALID;
  h->prof_on = on != 0;
  prof_reset(h);   // totals returned by get_profile accumulate from here
  return NERFHIP_OK;
}

int nerfhip_get_profile(nerfhip_t* h, nerfhip_prof* out) {
  int rc = check_handle(h); if (rc) return rc;
  if (!out) return fail(h, NERFHIP_E_INVALID, "null argument");
  HIPCHK(h, hipStreamSynchronize(h->stream));
  memset(out, 0, sizeof(*out));
  float acc[P_NCLASS] = {0};
  for (auto& r : h->prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) acc[r.cls] += ms;
  }
  out->mlp_fwd_ms = acc[P_MLP_FWD]; out->mlp_bwd_ms = acc[P_MLP_BWD]; out->mlp_wgrad_ms = acc[P_WGRAD];
  out->composite_ms = acc[P_COMPOSITE]; out->sampling_ms = acc[P_SAMPLING]; out->other_ms = acc[P_OTHER];
  out->comm_ms = acc[P_COMM];
  out->mlp_fwd_launches = h->fwd_launches;
  out->mlp_fwd_flop = h->flop_fwd; out->mlp_bwd_flop = h->flop_bwd; out->mlp_wgrad_flop = h->flop_wgrad;
  return NERFHIP_OK;
}

}  // extern "C"
