// Version / error plumbing, the RNG materialisation hooks and the MC epilogue.
#include <string.h>

#include "bt_api_internal.h"

namespace bt {

static thread_local char g_err[512] = "";

int set_error(int code, const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

// out[s][r][c][t] (natural order) <- the tap-major eps stream the fused kernels consume (bt_hip.h).
// One thread per Philox block: 4 consecutive values of e = (r*taps + t)*inner + c.
__global__ __launch_bounds__(256) void rng_normal_fill_kernel(RngKey k, const uint32_t* call_base, uint32_t sample0, long long rows,
                                                              long long inner, long long taps, float* __restrict__ out) {
  if (call_base) k.call += *call_base;
  const long long n = rows * inner * taps;
  const long long nq = (n + 3) >> 2;
  const int s = blockIdx.y;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
    float z[4];
    philox_normal4(k, sample0 + s, (uint32_t)q, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long e = q * 4 + j;
      if (e < n) {
        const long long c = e % inner, rt = e / inner;
        const long long t = rt % taps, r = rt / taps;
        out[(long long)s * n + (r * inner + c) * taps + t] = z[j];
      }
    }
  }
}

__global__ __launch_bounds__(256) void rng_sign_fill_kernel(RngKey k, const uint32_t* call_base, uint32_t sample0, long long n, float* __restrict__ out) {
  if (call_base) k.call += *call_base;
  const int s = blockIdx.y;
  const uint32_t key = sign_stream_key(k, sample0 + s);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[(long long)s * n + i] = hash_sign(key, (uint32_t)i);
}

__global__ __launch_bounds__(256) void softplus_kernel(const float* __restrict__ rho, float* __restrict__ sigma, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) sigma[i] = softplus(rho[i]);
}

// One block per batch row b: for each sample softmax over C classes, accumulate probabilities,
// entropies and raw logits. C <= 4096 handled by striding; S loop is sequential (fixed order).
__global__ __launch_bounds__(256) void mc_epilogue_kernel(int S, int B, int C, const float* __restrict__ logits, float* __restrict__ packed) {
  __shared__ float red[4];
  __shared__ float bcast;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  float* psum = packed + (long long)b * C;
  float* esum = packed + (long long)B * C + b;
  float* lsum = packed + (long long)B * C + B + (long long)b * C;
  float ent_total = 0.f;
  for (int c = t; c < C; c += 256) { psum[c] = 0.f; lsum[c] = 0.f; }
  for (int s = 0; s < S; ++s) {
    const float* row = logits + ((long long)s * B + b) * C;
    float mx = -INFINITY;
    for (int c = t; c < C; c += 256) mx = fmaxf(mx, row[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[w] = mx;
    __syncthreads();
    if (t == 0) bcast = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    mx = bcast;
    float den = 0.f;
    for (int c = t; c < C; c += 256) den += expf(row[c] - mx);
    for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o, 64);
    __syncthreads();
    if (lane == 0) red[w] = den;
    __syncthreads();
    if (t == 0) bcast = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    den = bcast;
    const float lden = logf(den);
    float ent = 0.f;
    for (int c = t; c < C; c += 256) {
      const float z = row[c] - mx;
      const float p = expf(z) / den;
      psum[c] += p;
      lsum[c] += row[c];
      ent -= (p > 0.f) ? p * (z - lden) : 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) ent += __shfl_xor(ent, o, 64);
    __syncthreads();
    if (lane == 0) red[w] = ent;
    __syncthreads();
    if (t == 0) ent_total += (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
  if (t == 0) *esum = ent_total;
}

}  // namespace bt

extern "C" int bt_version(void) { return BT_VERSION; }
extern "C" const char* bt_last_error_string(void) { return bt::g_err; }

extern "C" int bt_rng_normal_fill(const bt_rng* rng, uint32_t tensor_id, int32_t S, int64_t rows, int64_t inner, int64_t taps, float* out,
                                  bt_stream_t stream) {
  using namespace bt;
  if (!rng || !out || S <= 0 || rows <= 0 || inner <= 0 || taps <= 0 || tensor_id > 3 || S > 65535)
    return set_error(BT_ERR_BAD_ARG, "bt_rng_normal_fill: bad argument");
  const long long n = (long long)rows * inner * taps;
  if (n > (1ll << 34)) return set_error(BT_ERR_UNSUPPORTED, "bt_rng_normal_fill: tensor too large");
  const long long nq = (n + 3) >> 2;
  int gx = (int)((nq + 255) / 256);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(rng_normal_fill_kernel, dim3(gx, S), dim3(256), 0, (hipStream_t)stream, make_key(*rng, tensor_id), rng->call_base_dev,
                     rng->sample0, (long long)rows, (long long)inner, (long long)taps, out);
  return check_launch("bt_rng_normal_fill");
}

extern "C" int bt_rng_sign_fill(const bt_rng* rng, uint32_t tensor_id, int32_t S, int64_t n, float* out, bt_stream_t stream) {
  using namespace bt;
  if (!rng || !out || S <= 0 || n <= 0 || tensor_id > 3 || S > 65535 || n > 0xFFFFFFFFll) return set_error(BT_ERR_BAD_ARG, "bt_rng_sign_fill: bad argument");
  int gx = (int)((n + 255) / 256);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(rng_sign_fill_kernel, dim3(gx, S), dim3(256), 0, (hipStream_t)stream, make_key(*rng, tensor_id), rng->call_base_dev, rng->sample0, (long long)n, out);
  return check_launch("bt_rng_sign_fill");
}

extern "C" int bt_rng_philox_raw(uint64_t seed, const uint32_t ctr[4], uint32_t out_host[4]) {
  if (!ctr || !out_host) return bt::set_error(BT_ERR_BAD_ARG, "bt_rng_philox_raw: null argument");
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  bt::philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int i = 0; i < 4; ++i) out_host[i] = c[i];
  return BT_OK;
}

extern "C" int bt_mc_epilogue(int32_t S, int32_t B, int32_t C, const float* logits, float* packed, bt_stream_t stream) {
  using namespace bt;
  if (S <= 0 || B <= 0 || C <= 0 || !logits || !packed) return set_error(BT_ERR_BAD_ARG, "bt_mc_epilogue: bad argument");
  hipLaunchKernelGGL(mc_epilogue_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, S, B, C, logits, packed);
  return check_launch("bt_mc_epilogue");
}

extern "C" int bt_softplus(const float* rho, float* sigma, int64_t n, bt_stream_t stream) {
  using namespace bt;
  if (!rho || !sigma || n <= 0) return set_error(BT_ERR_BAD_ARG, "bt_softplus: bad argument");
  int gx = (int)((n + 255) / 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(softplus_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, rho, sigma, (long long)n);
  return check_launch("bt_softplus");
}
