/*
 * bt_hip.h -- C ABI of libbtorch_hip.so: the MI355X (gfx950) implementation of
 * bayesian-torch's stochastic variational-layer forward + KL hot path.
 *
 * The reference (godhj93/bayesian-torch) has no FFI of its own: the path is a
 * chain of ATen calls inside four nn.Module.forward() methods.  Each entry point
 * below replaces one such chain; the Python layer classes in
 * bayesian_torch_amd/layers/ bind them with ctypes (INTEGRATION.md shows the
 * stub a maintainer of the reference would add).  Citations are relative to
 * /root/reference/bayesian_torch/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless it says "host";
 *   - no allocation, no host synchronisation, no global state inside a call
 *     (graph-capturable); work is enqueued on `stream` (a hipStream_t);
 *   - return 0 on success, <0 on error (BT_ERR_*); bt_last_error_string() gives
 *     the text for the calling thread;
 *   - `S` MC samples are computed per call: out is [S][...]; sample s uses global
 *     sample id rng.sample0 + s, so results do not depend on how samples are
 *     split over calls, streams or ranks;
 *   - draws: when an eps_* / sign_* pointer is non-NULL the kernel READS the draw
 *     (parity mode: the caller supplies what torch's generator produced); when it
 *     is NULL the kernel generates it on chip from `rng` (Philox4x32-10 + Box-Muller
 *     for eps, a Philox-keyed integer hash for the Flipout signs) and nothing
 *     weight-sized or activation-sized is written to HBM besides `out`;
 *   - kl_out: NULL to skip; otherwise receives kl_weight + kl_bias of the layer
 *     (a function of the parameters only -- computed once, not per sample);
 *   - workspace: BT_WORKSPACE_BYTES device bytes per in-flight call, zero-filled
 *     once by the caller; calls leave it zeroed.
 */
#ifndef BT_HIP_H
#define BT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BT_VERSION 302
#define BT_WORKSPACE_BYTES 65536

#define BT_OK 0
#define BT_ERR_BAD_ARG (-1)
#define BT_ERR_UNSUPPORTED (-2)
#define BT_ERR_WORKSPACE (-3)
#define BT_ERR_HIP_BASE (-1000) /* -(1000 + hipError_t) */

typedef void *bt_stream_t; /* hipStream_t */

/* Counter-based RNG coordinates; used only for draws whose pointer is NULL. */
typedef struct bt_rng {
  uint64_t seed;                 /* Philox key */
  const uint32_t *call_base_dev; /* optional DEVICE word added to `call` when the kernel runs: lets a captured
                                    hipGraph draw fresh eps on every replay (advance the word between replays) */
  uint32_t call;                 /* advance once per forward call (a fresh draw per call) */
  uint32_t layer_id;             /* < 2^28: distinct per Bayesian layer of a model */
  uint32_t sample0;              /* global id of this call's first MC sample */
  uint32_t reserved;
} bt_rng;

/* Variational parameters and priors of one layer. weight is [Co][K] row-major
 * (Linear: K = in_features; Conv2d: K = (Ci/groups)*kh*kw, i.e. the native
 * [Co][Ci/groups][kh][kw] tensor).  Priors are full per-element tensors, read at
 * call time (users overwrite them: utils/util.py:102-117); only read when
 * kl_out != NULL. */
typedef struct bt_params {
  const float *mu_w, *rho_w;
  const float *mu_b, *rho_b; /* [Co] or both NULL */
  const float *prior_mu_w, *prior_sigma_w;
  const float *prior_mu_b, *prior_sigma_b;
  /* optional tap-major re-layout of the parameters, built by bt_pack_params and valid until mu_w / rho_w change:
   * mu_packed[(co*T + tap)*Ci4 + ci] = mu_w[co][ci][tap], sigma_packed[...] = log1p(exp(rho_w[co][ci][tap])),
   * Ci4 = Ci/groups rounded up to a multiple of 4 (padding holds 0). With them a weight unit (4 channels of one tap) is one
   * 16-byte load at the byte offset of its draw index, pruned taps are never touched in memory, and the kernel does no
   * softplus. Both NULL: the general kernel reads the natural layout. */
  const float *mu_packed, *sigma_packed;
  int32_t prior_kind; /* BT_PRIOR_NORMAL (kl_div 'normal'), BT_PRIOR_LAPLACE (kl_div 'laplace': base_variational_layer.py:74-97) */
  int32_t reserved;
} bt_params;
#define BT_PRIOR_NORMAL 0
#define BT_PRIOR_LAPLACE 1

/* Injected draws (NULL => generate on chip). Layouts: eps_w [S][Co*K],
 * eps_b [S][Co], sign_in [S][elements of one sample's x], sign_out
 * [S][elements of one sample's out]; sign values are the fp32 +1/-1/0 that
 * torch's uniform_(-1,1).sign() yields. */
typedef struct bt_draws {
  const float *eps_w, *eps_b;
  const float *sign_in, *sign_out; /* Flipout only */
  bt_rng rng;
} bt_draws;

typedef struct bt_conv2d_geom {
  int32_t B, Ci, H, W;    /* input  [B][Ci][H][W]  (NCHW) */
  int32_t Co, kh, kw;     /* weight [Co][Ci/groups][kh][kw] */
  int32_t sh, sw, ph, pw, dh, dw, groups;
} bt_conv2d_geom;

/* Optional fused output stage (inference-time Conv+BN(+add)(+ReLU) folding, SURVEY.md section 8(f) rank 4;
 * the reference folds BN only in its quantised path, models/bnn_to_qbnn.py:174-196). Applied per output
 * element after the layer's own result v (bias and Flipout perturbation included):
 *     v = v * scale[co] + shift[co]   (when scale != NULL; BatchNorm in eval mode is exactly this affine map)
 *     v = v + residual_s[idx]         (when residual != NULL; same layout as one sample of out)
 *     v = max(v, 0)                   (when relu != 0)                                                      */
typedef struct bt_epilogue {
  const float *scale, *shift;      /* [Co] each, or both NULL */
  const float *residual;           /* NULL, or [S][elements of one sample's out] / shared when stride is 0 */
  int64_t residual_sample_stride;  /* elements between samples, or 0 */
  int32_t relu;
  int32_t pool;                    /* BT_POOL_NONE, or BT_POOL_MAX_3x3_S2_P1: conv2d entry points only (see below) */
} bt_epilogue;

/* Fused max-pool of the output stage's result -- the ResNet stem's conv -> (folded BN) -> ReLU -> MaxPool2d(3, 2, 1) as one
 * launch: out is [S][B][Co][Hp][Wp] with Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1, NaN-propagating like
 * torch.nn.functional.max_pool2d. No residual with it. The launch returns BT_ERR_UNSUPPORTED (nothing written) when it cannot
 * fuse -- tiles that do not hold whole output images (large feature maps), Flipout, injected draws, input rows that are not
 * 16-byte aligned (W % 4 != 0): pool separately then. */
#define BT_POOL_NONE 0
#define BT_POOL_MAX_3x3_S2_P1 1

int bt_version(void);
const char *bt_last_error_string(void);
/* Diagnostic: the kernel instance (template name and tile) the calling thread's last fused-forward launch selected --
 * lets a benchmark attribute per-launch times to kernel instances the way rocprofv3's kernel trace does. */
const char *bt_last_kernel_name(void);
/* Scratch the split-K flavour of the forwards wants for this geometry (layers whose output map is one pixel: Linear, 1x1-map
 * convolutions; 0 for every other layer). Optional: hand the forward a workspace of BT_WORKSPACE_BYTES + this many bytes -- the
 * head zero-filled as always, the rest uninitialised -- and it may split the contraction over K-slices that meet in the scratch (a
 * fixed-order, deterministic combine); with a plain BT_WORKSPACE_BYTES workspace the other kernels serve the launch. */
size_t bt_fused_scratch_bytes(const bt_conv2d_geom *g, int32_t S);

/* ... and that launch's tile geometry, for a benchmark's roofline model: out[0..n) <- {workgroups, m_tiles, n_tiles, S, images per
 * tile, output rows per tile, output columns per tile, pixel_major, row_tiles, kl_slices, groups, n_bt, n_rt, n_ct, fused KL, 0}. */
int bt_last_launch_info(int64_t *out, int32_t n);

/* a5: LinearReparameterization.forward  (layers/variational_layers/linear_variational.py:160-181)
 *   out[s][b][o] = sum_k x_s[b][k] * (mu_w + log1p(exp(rho_w)) * eps_w[s])[o][k] + (mu_b + log1p(exp(rho_b)) * eps_b[s])[o]
 * x_sample_stride: 0 when all S samples share x [B][In]; else elements between samples (B*In). */
int bt_reparam_linear_fwd(int32_t B, int32_t In, int32_t Out, int32_t S,
                          const float *x, int64_t x_sample_stride,
                          const bt_params *p, const bt_draws *d, const bt_epilogue *ep /* or NULL */,
                          float *out /* [S][B][Out] */, float *kl_out /* [1] or NULL */,
                          void *workspace, size_t workspace_bytes, bt_stream_t stream);

/* a8: Conv2dReparameterization.forward  (layers/variational_layers/conv_variational.py:362-385)
 *   F.conv2d(x_s, mu + log1p(exp(rho)) * eps[s], bias_s, stride, padding, dilation, groups) as an implicit GEMM. */
int bt_reparam_conv2d_fwd(const bt_conv2d_geom *g, int32_t S,
                          const float *x, int64_t x_sample_stride,
                          const bt_params *p, const bt_draws *d, const bt_epilogue *ep /* or NULL */,
                          float *out /* [S][B][Co][Ho][Wo] */, float *kl_out,
                          void *workspace, size_t workspace_bytes, bt_stream_t stream);

/* a9: LinearFlipout.forward  (layers/flipout_layers/linear_flipout.py:145-174)
 *   out = x W_mu^T + mu_b + ((x o s_in) (sigma o eps)^T + sigma_b o eps_b) o s_out  -- both contractions share one x tile. */
int bt_flipout_linear_fwd(int32_t B, int32_t In, int32_t Out, int32_t S,
                          const float *x, int64_t x_sample_stride,
                          const bt_params *p, const bt_draws *d, const bt_epilogue *ep /* or NULL */,
                          float *out, float *kl_out,
                          void *workspace, size_t workspace_bytes, bt_stream_t stream);

/* a10: Conv2dFlipout.forward  (layers/flipout_layers/conv_flipout.py:370-417) */
int bt_flipout_conv2d_fwd(const bt_conv2d_geom *g, int32_t S,
                          const float *x, int64_t x_sample_stride,
                          const bt_params *p, const bt_draws *d, const bt_epilogue *ep /* or NULL */,
                          float *out, float *kl_out,
                          void *workspace, size_t workspace_bytes, bt_stream_t stream);

/* a1/a6/a12: kl_div 'normal' branch + kl_loss() + get_kl_loss()
 * (layers/base_variational_layer.py:68-72, linear_variational.py:146-158, models/dnn_to_bnn.py:157-165)
 *   kl_out[0] = sum_seg mean_i( log sp - log sq + (sq^2 + (mq - mp)^2) / (2 sp^2) - 0.5 ),  sq = log1p(exp(rho))
 * one launch over up to BT_KL_MAX_SEGMENTS tensors (a layer's weight and bias are two segments;
 * a whole model's layers can go in one call). Segment arrays are HOST arrays. */
#define BT_KL_MAX_SEGMENTS 64
#define BT_KL_RHO_IS_SIGMA 1u /* flags: the `rho` arrays already hold sigma (kl_div() called directly) */
#define BT_KL_PRIOR_LAPLACE 2u /* flags: kl_div's 'laplace' branch (base_variational_layer.py:74-97) for every segment:
                                  mean_i( log 2 - 0.5 log(2 pi sq^2) - 0.5 + E|w_i| ) against Laplace(0, 1) -- the reference
                                  hard-codes that prior and ignores the prior tensors; so does this flag (pass any valid tensors) */
int bt_kl_normal(int32_t n_segments, const float *const *mu, const float *const *rho,
                 const float *const *prior_mu, const float *const *prior_sigma, const int64_t *numel,
                 const int32_t *layer_of_segment /* host, non-decreasing, or NULL: the fp32 sum is formed as
                                                    sum_layers( sum_{segments of layer} mean ) like get_kl_loss */,
                 uint32_t flags, float *kl_out /* [1] */, void *workspace, size_t workspace_bytes, bt_stream_t stream);

/* Fills bt_params.mu_packed / sigma_packed (each Co * taps * Ci4 floats) from the natural [Co][Ci][taps] tensors; sigma uses
 * the kernels' own softplus, so a forward with or without the packed copies is bit-identical. */
int bt_pack_params(const float *mu_w, const float *rho_w, int64_t Co, int64_t Ci, int64_t taps,
                   float *mu_packed, float *sigma_packed, bt_stream_t stream);

/* Keeps the packed copies of up to BT_PACK_MAX_SEGMENTS layers in step with their parameters, checked ON THE DEVICE in the
 * stream (two launches, no host synchronisation, graph-capturable): a 64-bit fingerprint of every layer's natural-layout
 * (mu_w, rho_w) -- the wrapping sum of a 64-bit mix of (element index, mu bits, rho bits): order-independent, so deterministic --
 * is compared with the one stored in `state` when the pack was last built, and exactly the layers that differ (or carry
 * `force`) are re-packed in place. The reference's own code writes parameters through `.data` (models/dnn_to_bnn.py:65-71,95-101,
 * utils/util.py:102-117 in MOPED()), which no host-side version counter can see: with this call in front of the forward a
 * stale pack cannot be read. Segment array: HOST. */
#define BT_PACK_MAX_SEGMENTS 64
typedef struct bt_pack_seg {
  const float *mu_w, *rho_w;       /* the parameters, natural layout: what the fingerprint covers */
  const float *src_mu, *src_rho;   /* optional: an exact re-arrangement of them to pack from ([Co][Ci][taps]; the transposed-convolution
                                      layers pack their channel-transposed, flipped kernel); both NULL = mu_w / rho_w */
  float *mu_packed, *sigma_packed; /* [Co][taps][Ci4], rewritten in place when the fingerprint differs */
  uint64_t *state;                 /* DEVICE, 4 words owned by the layer, zero-filled once: accumulator (left zero), fingerprint
                                      of the packed copy, dirty flag of the last call, number of rebuilds so far */
  int64_t Co, Ci, taps;            /* geometry of the pack source */
  int32_t force;                   /* rebuild whatever the fingerprint says (first use of fresh buffers) */
  int32_t reserved;
} bt_pack_seg;
int bt_pack_sync(int32_t n_segments, const bt_pack_seg *segs, void *workspace, size_t workspace_bytes, bt_stream_t stream);

/* The on-chip draws, materialised (test / replay hook: the fused kernels never call these).
 * They emit exactly the streams the fused kernels consume for (rng, tensor_id):
 * tensor_id 0 = eps_w, 1 = eps_b, 2 = sign_in, 3 = sign_out.
 * Stream definition. eps element (row r, inner index c, tap t) of a [rows][inner][taps] tensor (a conv kernel
 * [Co][Ci/g][kh*kw]; Linear and bias: taps = 1) is normal number (c & 3) of Philox block (e >> 2) with
 * e = (r*taps + t)*inner4 + c,  inner4 = inner rounded up to a multiple of 4  -- tap-major and quad-aligned, so the
 * 4 values of one Philox block are 4 consecutive input channels of ONE tap: a kernel that skips taps which only ever
 * meet zero padding skips their RNG as well, and no block straddles two taps.
 * out is [S][rows*inner*taps] in the tensor's natural memory order. Signs: element i of the flat tensor. */
int bt_rng_normal_fill(const bt_rng *rng, uint32_t tensor_id, int32_t S, int64_t rows, int64_t inner, int64_t taps,
                       float *out, bt_stream_t stream);
int bt_rng_sign_fill(const bt_rng *rng, uint32_t tensor_id, int32_t S, int64_t n, float *out, bt_stream_t stream);
int bt_rng_philox_raw(uint64_t seed, const uint32_t ctr[4], uint32_t out_host[4]); /* host-side Philox4x32-10 KAT hook */

/* Backward of the four fused forwards (the reference relies on torch autograd of
 * linear_variational.py:163-181, conv_variational.py:366-385, linear_flipout.py:149-174, conv_flipout.py:376-417).
 * grad_out is dL/dout [S][B][Co][Ho][Wo] of a forward made with the same (geometry, S, p, d): the draws are REGENERATED on chip
 * from d->rng (or read, when d carries injected draws) -- nothing weight-sized is written to HBM.
 *   dx      [S][B][Ci][H][W] or NULL: per-sample input gradient (sum over S yourself when the forward shared x);
 *   dmu_w / drho_w  the parameters' own layout [Co][Ci/groups][kh][kw], both or neither:
 *           dmu = sum_s dW_s,  drho = sigmoid(rho) * sum_s eps_s o dW_s  (Flipout: mean path -> mu, perturbation path -> rho);
 *   workspace: bt_conv2d_bwd_workspace(g, S) bytes (partials of wgrad's sample / reduction groups and of dgrad's pieces of the
 *           output-channel reduction, summed in a fixed order; contents need not be initialised; needed whenever dx or dmu_w is).
 * Linear layers: g = {B, In, 1, 1, Out, 1, 1, 1, 1, 0, 0, 1, 1, 1}. p needs rho_w, mu_packed, sigma_packed. Bias gradients are
 * row sums of grad_out (not part of this call). */
size_t bt_conv2d_bwd_workspace(const bt_conv2d_geom *g, int32_t S);
int bt_conv2d_bwd(const bt_conv2d_geom *g, int32_t S, int32_t flipout, const float *x, int64_t x_sample_stride, const float *grad_out,
                  const bt_params *p, const bt_draws *d, float *dx, float *dmu_w, float *drho_w,
                  void *workspace, size_t workspace_bytes, bt_stream_t stream);
/* bt_conv2d_bwd with the layer's KL term differentiated in the same pass (a training step's loss holds both: the reference adds
 * get_kl_loss(model) / batch to the criterion, examples/main_bayesian_cifar_dnn2bnn.py:402-420): grad_kl = device scalar, the
 * upstream gradient of THIS layer's weight-KL mean (base_variational_layer.py:70-72; p->prior_kind selects :74-97). Then
 *   dmu_w[i] += grad_kl[0] / n * dKL_i/dmu,  drho_w[i] += grad_kl[0] / n * dKL_i/drho   (n = elements of mu_w)
 * in the op order autograd would add bt_kl_normal_bwd's tensors to bt_conv2d_bwd's: same bits, no second sweep, no KL-gradient
 * tensors. p additionally needs mu_w and (normal prior) prior_mu_w / prior_sigma_w. grad_kl NULL = bt_conv2d_bwd. Bias KL
 * gradients (bias-sized) stay with the caller. */
int bt_conv2d_bwd_kl(const bt_conv2d_geom *g, int32_t S, int32_t flipout, const float *x, int64_t x_sample_stride, const float *grad_out,
                     const bt_params *p, const bt_draws *d, const float *grad_kl, float *dx, float *dmu_w, float *drho_w,
                     void *workspace, size_t workspace_bytes, bt_stream_t stream);
/* Gradient of bt_kl_normal's mean over ONE tensor: dmu[i], drho[i] = d kl / d(mu_i, rho_i) * grad_kl[0] (grad_kl: device
 * scalar); flags: BT_KL_PRIOR_LAPLACE for the 'laplace' branch (priors unused then). */
int bt_kl_normal_bwd(const float *mu, const float *rho, const float *prior_mu, const float *prior_sigma, const float *grad_kl,
                     int64_t numel, uint32_t flags, float *dmu, float *drho, bt_stream_t stream);
/* The same for up to BT_KL_MAX_SEGMENTS tensors in ONE launch (the backward of a whole model's get_kl_loss): segment arrays are
 * HOST arrays like bt_kl_normal's; dmu[i] / drho[i] have numel[i] elements. */
int bt_kl_normal_bwd_segs(int32_t n_segments, const float *const *mu, const float *const *rho, const float *const *prior_mu,
                          const float *const *prior_sigma, const int64_t *numel, const float *grad_kl, uint32_t flags,
                          float *const *dmu, float *const *drho, bt_stream_t stream);

/* Contraction arithmetic of the fused forwards (process-wide; default from env BT_CONTRACTION = f32 | bf16x3 | bf16x2):
 *   0  automatic: wherever a launch is eligible, every fp32 operand is cut into three bf16 pieces (an EXACT split of the
 *      24-bit significand) and the product runs as the 6 piece products of weight >= 2^-16 on the bf16 matrix pipe with
 *      fp32 accumulation -- fp32-level accuracy at 6/16 of the fp32-MFMA time; other launches use fp32 MFMA;
 *   1  fp32 MFMA everywhere (the bit-exact fp32 FMA chain of round 1);
 *   2  two pieces, 3 product terms (relative error ~1e-5 per product): opt-in. */
int bt_set_contraction(int mode);

/* MC epilogue (examples/main_bayesian_cifar_dnn2bnn.py:551-557 and :402-412): from logits [S][B][C]
 * accumulate into packed[B*C + B + B*C] = [sum_s softmax | sum_s entropy | sum_s logits] (overwrites). */
int bt_mc_epilogue(int32_t S, int32_t B, int32_t C, const float *logits, float *packed, bt_stream_t stream);
/* MaxPool2d(kernel 3, stride 2, padding 1) over `planes` contiguous H x W planes (NCHW with planes = N * C): out holds planes x
 * ((H-1)/2+1) x ((W-1)/2+1). The pooling pass behind a stem whose fused launch returned BT_ERR_UNSUPPORTED for bt_epilogue.pool
 * (its tiles do not hold whole output images: the 112 x 112 ImageNet stem) -- `nn.MaxPool2d(3, 2, 1)` of the example models
 * (models/deterministic/resnet_large.py:120). Same comparisons as torch's max_pool2d (NaNs propagate): the same bits. */
int bt_maxpool_3x3s2(const float *in, float *out, int64_t planes, int32_t H, int32_t W, bt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BT_HIP_H */
