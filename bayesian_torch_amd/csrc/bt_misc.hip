// Version / error plumbing, the RNG materialisation hooks and the MC epilogue.
#include <string.h>

#include "bt_api_internal.h"

namespace bt {

static thread_local char g_err[512] = "";

static thread_local char g_kname[160] = "";
void note_kernel(const char* name) {
  strncpy(g_kname, name, sizeof(g_kname) - 1);
  g_kname[sizeof(g_kname) - 1] = 0;
}

int set_error(int code, const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

// out[s][r][c][t] (natural order) <- the tap-major, quad-aligned eps stream the fused kernels consume (bt_hip.h).
// One thread per Philox block: 4 consecutive channels of one (row, tap).
__global__ __launch_bounds__(256) void rng_normal_fill_kernel(RngKey k, const uint32_t* call_base, uint32_t sample0, long long rows,
                                                              long long inner, long long taps, float* __restrict__ out) {
  if (call_base) k.call += *call_base;
  const long long inner4 = (inner + 3) & ~3ll;
  const long long n = rows * inner * taps;
  const long long nq = rows * taps * (inner4 >> 2);
  const int s = blockIdx.y;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long)gridDim.x * 256) {
    float z[4];
    philox_normal4(k, sample0 + s, (uint32_t)q, z);
    const long long e = q * 4;
    const long long c0 = e % inner4, rt = e / inner4;
    const long long t = rt % taps, r = rt / taps;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < inner) out[(long long)s * n + (r * inner + c0 + j) * taps + t] = z[j];
  }
}

__global__ __launch_bounds__(256) void rng_sign_fill_kernel(RngKey k, const uint32_t* call_base, uint32_t sample0, long long n, float* __restrict__ out) {
  if (call_base) k.call += *call_base;
  const int s = blockIdx.y;
  const uint32_t key = sign_stream_key(k, sample0 + s);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[(long long)s * n + i] = hash_sign(key, (uint32_t)i);
}

// packed[(co*T + t)*C4 + c] <- natural[co][c][t]; one thread per packed element (reads are strided, done once per
// parameter update; writes are coalesced).
__global__ __launch_bounds__(256) void pack_params_kernel(const float* __restrict__ mu, const float* __restrict__ rho, long long Co,
                                                          long long C, long long T, float* __restrict__ mu_p, float* __restrict__ sg_p) {
  const long long C4 = (C + 3) & ~3ll, n = Co * T * C4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long c = i % C4, rt = i / C4;
    const long long t = rt % T, co = rt / T;
    float m = 0.f, sg = 0.f;
    if (c < C) {
      const long long src = (co * C + c) * T + t;
      m = mu[src];
      sg = softplus(rho[src]);
    }
    mu_p[i] = m;
    sg_p[i] = sg;
  }
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
extern "C" const char* bt_last_kernel_name(void) { return bt::g_kname; }
extern "C" const char* bt_last_error_string(void) { return bt::g_err; }

extern "C" int bt_rng_normal_fill(const bt_rng* rng, uint32_t tensor_id, int32_t S, int64_t rows, int64_t inner, int64_t taps, float* out,
                                  bt_stream_t stream) {
  using namespace bt;
  if (!rng || !out || S <= 0 || rows <= 0 || inner <= 0 || taps <= 0 || tensor_id > 3 || S > 65535)
    return set_error(BT_ERR_BAD_ARG, "bt_rng_normal_fill: bad argument");
  const long long nq = (long long)rows * taps * (((long long)inner + 3) >> 2);
  if (nq > (1ll << 32)) return set_error(BT_ERR_UNSUPPORTED, "bt_rng_normal_fill: tensor too large");
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

// C <= 64 (the usual classifier head): one WAVE per batch row, lane l owns sample 64 * chunk + l and walks the classes
// itself (no cross-lane work for the softmax); per-class sums over the samples are butterfly reductions, fixed order.
__global__ __launch_bounds__(256) void mc_epilogue_small_kernel(int S, int B, int C, const float* __restrict__ logits, float* __restrict__ packed) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float psum = 0.f, lsum = 0.f, ent_total = 0.f;  // lane c accumulates class c
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool on = s < S;
    const float* row = logits + ((long long)(on ? s : 0) * B + b) * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(row[c] - mx);
    const float lden = logf(den);
    float ent = 0.f;
    for (int c = 0; c < C; ++c) {
      const float x = row[c], z = x - mx, p = expf(z) / den;
      ent -= (p > 0.f) ? p * (z - lden) : 0.f;
      float ps = on ? p : 0.f, ls = on ? x : 0.f;
      for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o, 64), ls += __shfl_xor(ls, o, 64);
      if (lane == c) psum += ps, lsum += ls;
    }
    ent = on ? ent : 0.f;
    for (int o = 32; o > 0; o >>= 1) ent += __shfl_xor(ent, o, 64);
    ent_total += ent;
  }
  if (lane < C) {
    packed[(long long)b * C + lane] = psum;
    packed[(long long)B * C + B + (long long)b * C + lane] = lsum;
  }
  if (lane == 0) packed[(long long)B * C + b] = ent_total;
}

extern "C" int bt_mc_epilogue(int32_t S, int32_t B, int32_t C, const float* logits, float* packed, bt_stream_t stream) {
  using namespace bt;
  if (S <= 0 || B <= 0 || C <= 0 || !logits || !packed) return set_error(BT_ERR_BAD_ARG, "bt_mc_epilogue: bad argument");
  if (C <= 64)
    hipLaunchKernelGGL(mc_epilogue_small_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, B, C, logits, packed);
  else
    hipLaunchKernelGGL(mc_epilogue_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, S, B, C, logits, packed);
  return check_launch("bt_mc_epilogue");
}

// MaxPool2d(3, 2, 1) over NCHW planes -- the stem's pooling when the fused launch could not hold whole output images in a tile (the
// 112 x 112 ImageNet stem runs in bands of rows) and the output stage's pool therefore runs as a pass of its own. A thread = TWO
// neighbouring outputs of one row: per window row ONE 16-byte load (input columns 4 q .. 4 q + 3) and one 4-byte load (column 4 q - 1,
// a line its neighbour just fetched) instead of six scalar ones. Comparisons as torch's max_pool2d ((v > m) || isnan(v): NaNs
// propagate), so the result is the same bits. HBM-bound: ResNet50 / b256 / 16 samples reads 13.2 GB and writes 3.3 GB.
namespace bt {
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, long long items, int H, int W, int Ho, int Wo, int Wq) {
  auto nmax = [](float m, float v) { return (v > m || v != v) ? v : m; };
  const bool quads = (W & 3) == 0 && ((((uintptr_t)in) & 15u) == 0);   // rows of whole 16-byte column quads
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
    const long long r = it / Wq;             // output row (plane * Ho + py)
    const int q = (int)(it - r * Wq);        // outputs 2 q, 2 q + 1: input columns 4 q - 1 .. 4 q + 3
    const long long plane = r / Ho;
    const int py = (int)(r - plane * Ho);
    const float* const ip = in + plane * (long long)H * W;
    const int x0 = 4 * q;
    float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int h = 2 * py - 1 + k;
      if (h < 0 || h >= H) continue;
      const float* const row = ip + (long long)h * W;
      float c[5];   // columns x0 - 1 .. x0 + 3
      c[0] = x0 - 1 >= 0 ? row[x0 - 1] : -INFINITY;
      if (quads && x0 + 3 < W) {
        const float4 v = *reinterpret_cast<const float4*>(row + x0);
        c[1] = v.x, c[2] = v.y, c[3] = v.z, c[4] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[1 + j] = x0 + j < W ? row[x0 + j] : -INFINITY;
      }
      m0 = nmax(m0, c[0]), m0 = nmax(m0, c[1]), m0 = nmax(m0, c[2]);
      m1 = nmax(m1, c[2]), m1 = nmax(m1, c[3]), m1 = nmax(m1, c[4]);
    }
    float* const op = out + r * (long long)Wo + 2 * q;
    if (2 * q + 1 < Wo && (Wo & 1) == 0 && ((((uintptr_t)out) & 7u) == 0)) {
      *reinterpret_cast<float2*>(op) = make_float2(m0, m1);
    } else {
      op[0] = m0;
      if (2 * q + 1 < Wo) op[1] = m1;
    }
  }
}
}  // namespace bt

extern "C" int bt_maxpool_3x3s2(const float* in, float* out, int64_t planes, int32_t H, int32_t W, bt_stream_t stream) {
  using namespace bt;
  if (!in || !out || planes <= 0 || H <= 0 || W <= 0) return set_error(BT_ERR_BAD_ARG, "bt_maxpool_3x3s2: bad argument");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;   // floor((H + 2 - 3) / 2) + 1
  const int Wq = (Wo + 1) / 2;
  const long long items = (long long)planes * Ho * Wq;
  long long blocks = (items + 255) / 256;
  if (blocks > (1ll << 22)) blocks = 1ll << 22;   // grid-stride beyond that
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, items, (int)H, (int)W, Ho, Wo, Wq);
  return check_launch("bt_maxpool_3x3s2");
}

extern "C" int bt_pack_params(const float* mu_w, const float* rho_w, int64_t Co, int64_t Ci, int64_t taps, float* mu_packed,
                              float* sigma_packed, bt_stream_t stream) {
  using namespace bt;
  if (!mu_w || !rho_w || !mu_packed || !sigma_packed || Co <= 0 || Ci <= 0 || taps <= 0) return set_error(BT_ERR_BAD_ARG, "bt_pack_params: bad argument");
  const long long n = (long long)Co * taps * ((Ci + 3) & ~3ll);
  int gx = (int)((n + 255) / 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(pack_params_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, mu_w, rho_w, (long long)Co, (long long)Ci, (long long)taps,
                     mu_packed, sigma_packed);
  return check_launch("bt_pack_params");
}

// Test hook (not part of include/bt_hip.h): fills the whole LDS of every CU with NaN patterns (fp32 quiet NaNs = bf16 NaN pairs).
// LDS is not cleared between kernels, so a kernel that reads an LDS slot it never wrote sees them: tests/test_gpu_split.py runs
// the split kernels behind this and expects finite, unchanged results.
namespace bt {
__global__ __launch_bounds__(1024) void poison_lds_kernel(unsigned* sink) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = 0x7FC07FC0u;
  __syncthreads();
  if (lds[(threadIdx.x * 37) % (160 * 1024 / 4)] == 1u) sink[0] = 1u;   // (keeps the stores alive)
}
}  // namespace bt
extern "C" int bt_debug_poison_lds(void* scratch_word, bt_stream_t stream) {
  using namespace bt;
  static bool flags[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "bt_debug_poison_lds: hipGetDevice failed");
  if (!flags[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return set_error(BT_ERR_HIP_BASE, "bt_debug_poison_lds: cannot raise the dynamic LDS limit");
    flags[dev] = true;
  }
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned*)scratch_word);
  return check_launch("bt_debug_poison_lds");
}
