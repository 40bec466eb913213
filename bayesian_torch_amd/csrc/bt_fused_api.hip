// C-ABI entry points of the four fused forwards: argument validation, geometry, launch.
#include <string.h>

#include "bt_fused_fwd.h"

namespace bt {
int launch_reparam(bool linear, FwdArgs& a, hipStream_t stream);
int launch_reparam_inj(bool linear, FwdArgs& a, hipStream_t stream);
int launch_flipout(bool linear, FwdArgs& a, hipStream_t stream);
int launch_flipout_inj(bool linear, FwdArgs& a, hipStream_t stream);

static unsigned long long* g_dbg = nullptr;
static thread_local long long g_launch_info[16] = {};
static inline bool al16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

static int run(bool flip, bool linear, const bt_conv2d_geom& g, int S, const float* x, int64_t x_sample_stride, const bt_params* p,
               const bt_draws* d, const bt_epilogue* ep, float* out, float* kl_out, void* ws, size_t ws_bytes, bt_stream_t stream, const char* who) {
  char msg[256];
  auto bad = [&](const char* what) {
    snprintf(msg, sizeof(msg), "%s: %s", who, what);
    return set_error(BT_ERR_BAD_ARG, msg);
  };
  if (!x || !p || !d || !out) return bad("null argument");
  if (!p->mu_w || !p->rho_w) return bad("mu_w / rho_w are required");
  if ((p->mu_b == nullptr) != (p->rho_b == nullptr)) return bad("mu_b and rho_b must both be given or both be NULL");
  if (S <= 0) return bad("S must be >= 1");
  if (g.B <= 0 || g.Ci <= 0 || g.H <= 0 || g.W <= 0 || g.Co <= 0 || g.kh <= 0 || g.kw <= 0) return bad("non-positive dimension");
  if (g.sh <= 0 || g.sw <= 0 || g.dh <= 0 || g.dw <= 0 || g.ph < 0 || g.pw < 0 || g.groups <= 0) return bad("bad stride / padding / dilation / groups");
  if (g.Ci % g.groups || g.Co % g.groups) return bad("invalid in_channels size");  // conv_variational.py:270-273
  if (x_sample_stride < 0) return bad("negative x_sample_stride");
  if (kl_out) {
    if (!p->prior_mu_w || !p->prior_sigma_w) return bad("kl_out given but weight priors are NULL");
    if (p->mu_b && (!p->prior_mu_b || !p->prior_sigma_b)) return bad("kl_out given but bias priors are NULL");
    if (!ws || ws_bytes < BT_WORKSPACE_BYTES) {
      snprintf(msg, sizeof(msg), "%s: workspace smaller than BT_WORKSPACE_BYTES", who);
      return set_error(BT_ERR_WORKSPACE, msg);
    }
  }
  if (p->mu_b == nullptr && d->eps_b) return bad("eps_b given for a layer without bias");
  if (!flip && (d->sign_in || d->sign_out)) return bad("sign tensors are Flipout-only");
  if (ep && ((ep->scale == nullptr) != (ep->shift == nullptr))) return bad("epilogue scale and shift must both be given or both be NULL");
  if (ep && ep->residual_sample_stride < 0) return bad("negative residual_sample_stride");

  const int Ho = (g.H + 2 * g.ph - g.dh * (g.kh - 1) - 1) / g.sh + 1;
  const int Wo = (g.W + 2 * g.pw - g.dw * (g.kw - 1) - 1) / g.sw + 1;
  if (Ho <= 0 || Wo <= 0) return bad("empty output");

  FwdArgs a;
  FwdArgs zero = {}; a = zero;
  a.x = x, a.mu_w = p->mu_w, a.rho_w = p->rho_w, a.mu_b = p->mu_b, a.rho_b = p->rho_b;
  a.pmu_w = p->prior_mu_w, a.psig_w = p->prior_sigma_w, a.pmu_b = p->prior_mu_b, a.psig_b = p->prior_sigma_b;
  if ((p->mu_packed == nullptr) != (p->sigma_packed == nullptr)) return bad("mu_packed and sigma_packed must both be given or both be NULL");
  a.mu_pk = p->mu_packed, a.sig_pk = p->sigma_packed;
  a.eps_w = d->eps_w, a.eps_b = d->eps_b, a.sign_in = d->sign_in, a.sign_out = d->sign_out;
  a.out = out, a.kl_out = kl_out;
  a.slots = kl_out ? ws_slots(ws) : nullptr;
  a.counter = kl_out ? ws_counter(ws) : nullptr;
  // a workspace larger than BT_WORKSPACE_BYTES carries scratch for the split-K (skinny) flavour behind its zeroed head
  if (ws && ws_bytes > BT_WORKSPACE_BYTES) {
    a.sk_scratch = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + BT_WORKSPACE_BYTES);
    a.sk_scratch_bytes = (long long)(ws_bytes - BT_WORKSPACE_BYTES);
    a.sk_tickets = reinterpret_cast<unsigned*>(ws_slots(ws) + 4000);
  }
  a.B = g.B, a.Ci = g.Ci, a.H = g.H, a.W = g.W, a.Co = g.Co, a.KH = g.kh, a.KW = g.kw;
  a.SH = g.sh, a.SW = g.sw, a.PH = g.ph, a.PW = g.pw, a.DH = g.dh, a.DW = g.dw, a.G = g.groups;
  a.Ho = Ho, a.Wo = Wo, a.HoWo = Ho * Wo;
  const long long M = (long long)g.B * Ho * Wo;
  a.Cig = g.Ci / g.groups, a.Cog = g.Co / g.groups;
  const long long K = (long long)a.Cig * g.kh * g.kw;
  a.x_elems = (long long)g.B * g.Ci * g.H * g.W;
  a.out_elems = M * g.Co;
  a.w_elems = (long long)g.Co * K;
  // 32-bit index budget of the kernel (tile indices, hashed sign indices, Philox block index)
  if (M >= (1ll << 30) || K >= (1ll << 30) || a.x_elems >= (1ll << 30) || a.out_elems >= (1ll << 30) || a.w_elems >= (1ll << 30))
    return set_error(BT_ERR_UNSUPPORTED, "fused forward: a tensor of 2^30 elements or more exceeds the kernel's 32-bit offsets");
  a.M = (int)M, a.K = (int)K, a.S = S;
  a.T = g.kh * g.kw, a.HW = g.H * g.W;
  if (a.T > kMaxTaps) return set_error(BT_ERR_UNSUPPORTED, "fused forward: kernels larger than 128 taps are not supported");
  // pixel-major tiles prune padding taps per output pixel; worth it when images are tiny (2x2 outputs: 4 of 9 taps)
  a.pixel_major = (!linear && a.HoWo >= 2 && a.HoWo <= 4 && (g.ph > 0 || g.pw > 0)) ? 1 : 0;
  a.x_sample_stride = x_sample_stride;
  a.w_vec = linear && ((K & 3) == 0) && al16(p->mu_w) && al16(p->rho_w) && (!d->eps_w || al16(d->eps_w));
  a.x_vec = linear && ((K & 3) == 0) && al16(x) && ((x_sample_stride & 3) == 0) && (!d->sign_in || al16(d->sign_in));
  // The fused KL sweep is the Gaussian closed form. A Laplace-prior layer gets its KL from the standalone kernel, enqueued
  // right behind the forward on the same stream (same result slot, same workspace).
  if (p->prior_kind != BT_PRIOR_NORMAL && p->prior_kind != BT_PRIOR_LAPLACE) return bad("unknown prior_kind");
  const bool kl_after = kl_out && p->prior_kind == BT_PRIOR_LAPLACE;
  a.do_kl = kl_out != nullptr && !kl_after;
  if (kl_after) a.kl_out = nullptr;
  a.seed_lo = (uint32_t)d->rng.seed, a.seed_hi = (uint32_t)(d->rng.seed >> 32);
  if (ep) a.ep_scale = ep->scale, a.ep_shift = ep->shift, a.ep_res = ep->residual, a.ep_res_stride = ep->residual_sample_stride, a.ep_relu = ep->relu;
  if (ep && ep->pool != BT_POOL_NONE) {
    if (ep->pool != BT_POOL_MAX_3x3_S2_P1) return bad("unknown epilogue pool mode");
    if (linear) return bad("the fused max-pool belongs to the conv2d entry points");
    if (ep->residual) return bad("the fused max-pool takes no residual");
    a.ep_pool = 1, a.ep_Hp = (Ho - 1) / 2 + 1, a.ep_Wp = (Wo - 1) / 2 + 1;
    a.out_elems = (long long)g.B * g.Co * a.ep_Hp * a.ep_Wp;
  }
  a.out_vec4 = (!linear && !a.pixel_major && a.HoWo > 1 && (a.Wo & 3) == 0 && al16(out) && (!a.ep_res || (al16(a.ep_res) && (a.ep_res_stride & 3) == 0)) &&
                (!d->sign_out || al16(d->sign_out))) ? 1 : 0;
  a.dbg = g_dbg;
  a.call = d->rng.call, a.call_base = d->rng.call_base_dev, a.layer_id = d->rng.layer_id, a.sample0 = d->rng.sample0;
  // draws are either all injected or all generated on chip (one compile-time flavour each)
  const bool inj = d->eps_w != nullptr;
  const bool all_inj = inj && (!p->mu_b || d->eps_b) && (!flip || (d->sign_in && d->sign_out));
  const bool none_inj = !d->eps_w && !d->eps_b && !d->sign_in && !d->sign_out;
  if (!all_inj && !none_inj) return bad("inject all draws of the layer (eps_w, eps_b when biased, both sign tensors for Flipout) or none");
  if (inj && a.ep_pool) return set_error(BT_ERR_UNSUPPORTED, "fused max-pool: not available with injected draws");
  int rc;
  if (inj) rc = flip ? launch_flipout_inj(linear, a, (hipStream_t)stream) : launch_reparam_inj(linear, a, (hipStream_t)stream);
  else rc = flip ? launch_flipout(linear, a, (hipStream_t)stream) : launch_reparam(linear, a, (hipStream_t)stream);
  if (rc == BT_OK) {   // tile geometry of the launch just made (bt_last_launch_info)
    const long long v[16] = {a.total_blocks, a.m_tiles, a.n_tiles, a.S, a.t_NI, a.t_R, a.t_Wt, a.pixel_major, a.row_taps, a.kl_slices, a.G, a.n_bt, a.n_rt, a.n_ct, a.do_kl, 0};
    for (int i = 0; i < 16; ++i) g_launch_info[i] = v[i];
  }
  if (rc == BT_OK && kl_after) {
    const float* mu[2] = {p->mu_w, p->mu_b};
    const float* rho[2] = {p->rho_w, p->rho_b};
    const float* pm[2] = {p->prior_mu_w, p->prior_mu_b};
    const float* ps[2] = {p->prior_sigma_w, p->prior_sigma_b};
    const int64_t n[2] = {a.w_elems, (int64_t)g.Co};
    const int32_t lay[2] = {0, 0};
    rc = bt_kl_normal(p->mu_b ? 2 : 1, mu, rho, pm, ps, n, lay, BT_KL_PRIOR_LAPLACE, kl_out, ws, ws_bytes, stream);
  }
  return rc;
}

static bt_conv2d_geom linear_geom(int B, int In, int Out) {
  bt_conv2d_geom g;
  g.B = B, g.Ci = In, g.H = 1, g.W = 1, g.Co = Out, g.kh = 1, g.kw = 1;
  g.sh = g.sw = 1, g.ph = g.pw = 0, g.dh = g.dw = 1, g.groups = 1;
  return g;
}
}  // namespace bt

extern "C" int bt_reparam_linear_fwd(int32_t B, int32_t In, int32_t Out, int32_t S, const float* x, int64_t x_sample_stride,
                                     const bt_params* p, const bt_draws* d, const bt_epilogue* ep, float* out, float* kl_out, void* ws, size_t ws_bytes,
                                     bt_stream_t stream) {
  return bt::run(false, true, bt::linear_geom(B, In, Out), S, x, x_sample_stride, p, d, ep, out, kl_out, ws, ws_bytes, stream, "bt_reparam_linear_fwd");
}
extern "C" int bt_flipout_linear_fwd(int32_t B, int32_t In, int32_t Out, int32_t S, const float* x, int64_t x_sample_stride,
                                     const bt_params* p, const bt_draws* d, const bt_epilogue* ep, float* out, float* kl_out, void* ws, size_t ws_bytes,
                                     bt_stream_t stream) {
  return bt::run(true, true, bt::linear_geom(B, In, Out), S, x, x_sample_stride, p, d, ep, out, kl_out, ws, ws_bytes, stream, "bt_flipout_linear_fwd");
}
extern "C" int bt_reparam_conv2d_fwd(const bt_conv2d_geom* g, int32_t S, const float* x, int64_t x_sample_stride, const bt_params* p,
                                     const bt_draws* d, const bt_epilogue* ep, float* out, float* kl_out, void* ws, size_t ws_bytes, bt_stream_t stream) {
  if (!g) return bt::set_error(BT_ERR_BAD_ARG, "bt_reparam_conv2d_fwd: null geometry");
  return bt::run(false, false, *g, S, x, x_sample_stride, p, d, ep, out, kl_out, ws, ws_bytes, stream, "bt_reparam_conv2d_fwd");
}
extern "C" int bt_flipout_conv2d_fwd(const bt_conv2d_geom* g, int32_t S, const float* x, int64_t x_sample_stride, const bt_params* p,
                                     const bt_draws* d, const bt_epilogue* ep, float* out, float* kl_out, void* ws, size_t ws_bytes, bt_stream_t stream) {
  if (!g) return bt::set_error(BT_ERR_BAD_ARG, "bt_flipout_conv2d_fwd: null geometry");
  return bt::run(true, false, *g, S, x, x_sample_stride, p, d, ep, out, kl_out, ws, ws_bytes, stream, "bt_flipout_conv2d_fwd");
}

namespace bt { long long skinny_scratch_bytes(const bt_conv2d_geom& g, int S); }
extern "C" size_t bt_fused_scratch_bytes(const bt_conv2d_geom* g, int32_t S) {
  if (!g || S <= 0) return 0;
  const long long n = bt::skinny_scratch_bytes(*g, S);
  return n > 0 ? (size_t)n : 0;
}

extern "C" int bt_last_launch_info(int64_t* out, int32_t n) {
  if (!out || n <= 0) return bt::set_error(BT_ERR_BAD_ARG, "bt_last_launch_info: bad argument");
  for (int i = 0; i < n; ++i) out[i] = i < 16 ? (int64_t)bt::g_launch_info[i] : 0;
  return BT_OK;
}

// Diagnostic hook (not part of include/bt_hip.h): device buffer of >= 256 u64 that block 0 of every fused launch
// fills with s_memtime stamps per stage (consumer wave 0: [2+2st, 3+2st]; producer wave 4: [128+2st, 129+2st]).
extern "C" void bt_debug_set_stamp_buffer(void* p) { bt::g_dbg = (unsigned long long*)p; }
