// K6: standalone closed-form KL of mean-field Gaussians against per-element Gaussian priors.
// Replaces kl_loss() / get_kl_loss() (reference layers/base_variational_layer.py:68-72,
// variational_layers/linear_variational.py:146-158, models/dnn_to_bnn.py:157-165): there it is
// ~12 full-tensor ATen passes per layer; here one pass, 16 B/element (mu, rho, prior_mu,
// prior_sigma), HBM-bound, for up to 64 tensors (a whole model) per launch.
#include "bt_api_internal.h"

namespace bt {

struct KlSegs {
  const float* mu[BT_KL_MAX_SEGMENTS];
  const float* rho[BT_KL_MAX_SEGMENTS];
  const float* pmu[BT_KL_MAX_SEGMENTS];
  const float* psig[BT_KL_MAX_SEGMENTS];
  long long n[BT_KL_MAX_SEGMENTS];
  int first_block[BT_KL_MAX_SEGMENTS + 1];  // blocks [first_block[i], first_block[i+1]) work on segment i
  unsigned long long new_layer;             // bit i: segment i starts a new layer (fp32 summation grouping)
  int nseg;
  int rho_is_sigma;
  int laplace;  // 'laplace' branch of kl_div (prior tensors are ignored, as in the reference) instead of the Gaussian closed form
};

constexpr int kKlThreads = 256;
constexpr int kKlVecPerThread = 4;                                   // float4 loads in flight per tensor per thread
constexpr int kKlElemsPerBlock = kKlThreads * 4 * kKlVecPerThread;  // 4096 elements per block-iteration
constexpr int kKlMaxBlocksPerSeg = 1024;

__global__ __launch_bounds__(kKlThreads) void kl_normal_kernel(KlSegs sg, float* kl_out, double* slots, unsigned* counter,
                                                              int total_blocks) {
  __shared__ double red[4];
  __shared__ int is_last;
  // locate this block's segment (nseg <= 64: linear scan on scalars)
  int seg = 0;
  while (seg + 1 < sg.nseg && (int)blockIdx.x >= sg.first_block[seg + 1]) ++seg;
  const int nb = sg.first_block[seg + 1] - sg.first_block[seg];
  const int lb = blockIdx.x - sg.first_block[seg];
  const long long n = sg.n[seg];
  const float* __restrict__ mu = sg.mu[seg];
  const float* __restrict__ rho = sg.rho[seg];
  const float* __restrict__ pmu = sg.pmu[seg];
  const float* __restrict__ psig = sg.psig[seg];
  const bool vec_ok = ((((uintptr_t)mu | (uintptr_t)rho | (uintptr_t)pmu | (uintptr_t)psig) & 15u) == 0);

  const bool is_sigma = sg.rho_is_sigma != 0;
  auto sig = [&](float v) { return is_sigma ? v : softplus(v); };
  const bool lap = sg.laplace != 0;
  auto kl_term = [&](float mq, float sq, float mp, float sp) { return lap ? kl_term_laplace(mq, sq) : bt::kl_term(mq, sq, mp, sp); };
  double acc = 0.0;
  const long long n4 = vec_ok ? (n >> 2) : 0;  // float4 groups
  for (long long base = (long long)lb * (kKlThreads * kKlVecPerThread); base < n4; base += (long long)nb * (kKlThreads * kKlVecPerThread)) {
    float4 m[kKlVecPerThread], r[kKlVecPerThread], pm[kKlVecPerThread], ps[kKlVecPerThread];
#pragma unroll
    for (int v = 0; v < kKlVecPerThread; ++v) {
      const long long i = base + v * kKlThreads + threadIdx.x;
      if (i < n4) {
        m[v] = reinterpret_cast<const float4*>(mu)[i];
        r[v] = reinterpret_cast<const float4*>(rho)[i];
        pm[v] = reinterpret_cast<const float4*>(pmu)[i];
        ps[v] = reinterpret_cast<const float4*>(psig)[i];
      }
    }
#pragma unroll
    for (int v = 0; v < kKlVecPerThread; ++v) {
      const long long i = base + v * kKlThreads + threadIdx.x;
      if (i < n4) {
        float t = kl_term(m[v].x, sig(r[v].x), pm[v].x, ps[v].x);
        t += kl_term(m[v].y, sig(r[v].y), pm[v].y, ps[v].y);
        float t2 = kl_term(m[v].z, sig(r[v].z), pm[v].z, ps[v].z);
        t2 += kl_term(m[v].w, sig(r[v].w), pm[v].w, ps[v].w);
        acc += (double)t + (double)t2;
      }
    }
  }
  // tail (or everything, when a pointer is not 16-B aligned)
  for (long long i = (n4 << 2) + (long long)lb * kKlThreads + threadIdx.x; i < n; i += (long long)nb * kKlThreads)
    acc += (double)kl_term(mu[i], sig(rho[i]), pmu[i], psig[i]);

  const double bsum = block_sum_256(acc, red);
  if (threadIdx.x == 0) is_last = publish_and_ticket_wt(slots, counter, blockIdx.x, bsum, (unsigned)total_blocks) ? 1 : 0;
  __syncthreads();
  if (!is_last) return;
  // last arriver: fixed-ORDER finish, in parallel. A segment's block partials are summed by the 256 threads in a fixed tree (thread t
  // adds the slots t, t + 256, ... in increasing order; then the wave butterflies and the four wave sums in index order): the same
  // tree every time, so the result is deterministic -- and a whole model's ~2,700 slots no longer pass through ONE thread's serial
  // chain of agent-scope loads (round 3: that chain was 429 us of a 3.95 ms ResNet18 training step, and of every get_kl_loss()).
  // mean per segment in fp64 -> fp32, segments added in fp32 (the reference adds fp32 0-dim tensors: kl_weight + kl_bias, then += across layers).
  __shared__ double seg_sum[BT_KL_MAX_SEGMENTS];
  for (int s = 0; s < sg.nseg; ++s) {
    double t = 0.0;
    for (int b = sg.first_block[s] + (int)threadIdx.x; b < sg.first_block[s + 1]; b += kKlThreads) t += __hip_atomic_load(&slots[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = wave_sum(t);
    __syncthreads();   // (red is free again)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) seg_sum[s] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.0f, layer = 0.0f;
    for (int s = 0; s < sg.nseg; ++s) {
      const float mean = (float)(seg_sum[s] / (double)sg.n[s]);
      if ((sg.new_layer >> s) & 1ull) {
        if (s) total += layer;  // 0.0f + x is exact, so the first layer enters unrounded
        layer = mean;
      } else {
        layer += mean;
      }
    }
    kl_out[0] = total + layer;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // leave the workspace zeroed
  }
}

}  // namespace bt

extern "C" int bt_kl_normal(int32_t n_segments, const float* const* mu, const float* const* rho, const float* const* prior_mu,
                            const float* const* prior_sigma, const int64_t* numel, const int32_t* layer_of_segment, uint32_t flags,
                            float* kl_out, void* workspace, size_t workspace_bytes, bt_stream_t stream) {
  using namespace bt;
  if (n_segments <= 0 || n_segments > BT_KL_MAX_SEGMENTS) return set_error(BT_ERR_BAD_ARG, "bt_kl_normal: n_segments must be in [1, 64]");
  if (!mu || !rho || !prior_mu || !prior_sigma || !numel || !kl_out) return set_error(BT_ERR_BAD_ARG, "bt_kl_normal: null argument");
  if (!workspace || workspace_bytes < BT_WORKSPACE_BYTES) return set_error(BT_ERR_WORKSPACE, "bt_kl_normal: workspace smaller than BT_WORKSPACE_BYTES");
  KlSegs sg;
  int blocks = 0;
  for (int i = 0; i < n_segments; ++i) {
    if (numel[i] <= 0 || !mu[i] || !rho[i] || !prior_mu[i] || !prior_sigma[i]) return set_error(BT_ERR_BAD_ARG, "bt_kl_normal: empty or null segment");
    sg.mu[i] = mu[i];
    sg.rho[i] = rho[i];
    sg.pmu[i] = prior_mu[i];
    sg.psig[i] = prior_sigma[i];
    sg.n[i] = numel[i];
    long long nb = (numel[i] + kKlElemsPerBlock - 1) / kKlElemsPerBlock;
    if (nb > kKlMaxBlocksPerSeg) nb = kKlMaxBlocksPerSeg;
    sg.first_block[i] = blocks;
    blocks += (int)nb;
  }
  sg.first_block[n_segments] = blocks;
  sg.nseg = n_segments;
  sg.rho_is_sigma = (flags & BT_KL_RHO_IS_SIGMA) ? 1 : 0;
  sg.laplace = (flags & BT_KL_PRIOR_LAPLACE) ? 1 : 0;
  sg.new_layer = 0;
  for (int i = 0; i < n_segments; ++i) {
    if (layer_of_segment && i && layer_of_segment[i] < layer_of_segment[i - 1]) return set_error(BT_ERR_BAD_ARG, "bt_kl_normal: layer_of_segment must be non-decreasing");
    if (i == 0 || !layer_of_segment || layer_of_segment[i] != layer_of_segment[i - 1]) sg.new_layer |= 1ull << i;
  }
  if (blocks > kMaxSlots) {  // only possible with many huge segments: cap per-segment blocks harder
    return set_error(BT_ERR_UNSUPPORTED, "bt_kl_normal: too many blocks for the workspace; split the call");
  }
  hipLaunchKernelGGL(kl_normal_kernel, dim3(blocks), dim3(kKlThreads), 0, (hipStream_t)stream, sg, kl_out, ws_slots(workspace),
                     ws_counter(workspace), blocks);
  return check_launch("bt_kl_normal");
}
