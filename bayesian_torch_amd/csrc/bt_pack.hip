// bt_pack_sync: keep the tap-major (mu, softplus(rho)) copies the fast kernels read in step with the parameters WITHOUT trusting
// the host to notice a change.
//
// The reference's own idioms write parameters through `.data` (models/dnn_to_bnn.py:65-71,95-101 `mu_kernel.data.copy_`,
// utils/util.py:102-117 in MOPED()), which no host-side version counter sees. So the check is made on the device, every time, in
// the stream: launch 1 sweeps the natural-layout (mu, rho) of up to 64 layers (8 B/weight, HBM-bound: ResNet18's 89 MB ~ 25 us)
// into one 64-bit fingerprint per layer -- the wrapping sum over the elements of a 64-bit mix of (index, mu bits, rho bits);
// integer addition commutes, so the value does not depend on the order the blocks arrive in -- and its last block compares each
// layer's fingerprint with the one stored when its pack was last built. Launch 2 rebuilds the packs of exactly the layers that
// differ (its blocks return at once for the others). Nothing comes back to the host, so both launches sit inside a captured
// HIP graph like any other kernel: a replay follows parameter updates.
#include "bt_api_internal.h"

namespace bt {

struct PackSegs {
  const float* mu[BT_PACK_MAX_SEGMENTS];
  const float* rho[BT_PACK_MAX_SEGMENTS];
  const float* src_mu[BT_PACK_MAX_SEGMENTS];
  const float* src_rho[BT_PACK_MAX_SEGMENTS];
  float* mu_p[BT_PACK_MAX_SEGMENTS];
  float* sg_p[BT_PACK_MAX_SEGMENTS];
  unsigned long long* state[BT_PACK_MAX_SEGMENTS];
  long long n[BT_PACK_MAX_SEGMENTS];        // elements of the natural tensors
  long long C[BT_PACK_MAX_SEGMENTS], T[BT_PACK_MAX_SEGMENTS], np[BT_PACK_MAX_SEGMENTS];   // pack geometry: channels, taps, packed elements
  int first_block[BT_PACK_MAX_SEGMENTS + 1];
  unsigned long long force;                 // bit i: rebuild segment i whatever its fingerprint says
  int nseg;
};

constexpr int kFpThreads = 256;
constexpr int kFpElemsPerBlock = kFpThreads * 4 * 4;

__device__ __forceinline__ unsigned long long fp_mix(unsigned long long i, float m, float r) {
  unsigned long long h = (((unsigned long long)__float_as_uint(m)) << 32 | (unsigned long long)__float_as_uint(r)) + i * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return h;
}

__global__ __launch_bounds__(kFpThreads) void pack_fingerprint_kernel(PackSegs sg, unsigned* counter, int total_blocks) {
  __shared__ int is_last;
  int seg = 0;
  while (seg + 1 < sg.nseg && (int)blockIdx.x >= sg.first_block[seg + 1]) ++seg;
  const int nb = sg.first_block[seg + 1] - sg.first_block[seg], lb = blockIdx.x - sg.first_block[seg];
  const long long n = sg.n[seg];
  const float* __restrict__ mu = sg.mu[seg];
  const float* __restrict__ rho = sg.rho[seg];
  const bool vec_ok = ((((uintptr_t)mu | (uintptr_t)rho) & 15u) == 0);
  unsigned long long acc = 0ull;
  const long long n4 = vec_ok ? (n >> 2) : 0;
  for (long long base = (long long)lb * (kFpThreads * 4); base < n4; base += (long long)nb * (kFpThreads * 4)) {
    float4 m[4], r[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const long long i = base + v * kFpThreads + threadIdx.x;
      if (i < n4) m[v] = reinterpret_cast<const float4*>(mu)[i], r[v] = reinterpret_cast<const float4*>(rho)[i];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const long long i = base + v * kFpThreads + threadIdx.x;
      if (i < n4) {
        const unsigned long long e = (unsigned long long)i * 4ull;
        acc += fp_mix(e, m[v].x, r[v].x) + fp_mix(e + 1, m[v].y, r[v].y) + fp_mix(e + 2, m[v].z, r[v].z) + fp_mix(e + 3, m[v].w, r[v].w);
      }
    }
  }
  for (long long i = (n4 << 2) + (long long)lb * kFpThreads + threadIdx.x; i < n; i += (long long)nb * kFpThreads) acc += fp_mix((unsigned long long)i, mu[i], rho[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  // One agent-scope add per wave (performed at L2), drained before the block's ticket: the last arriver's agent-scope loads see
  // every add (MI355X_MICROARCH.md "Valid forms": all handed-off words written and read with agent-scope atomics).
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&sg.state[seg][0], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) is_last = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)total_blocks - 1u) ? 1 : 0;
  __syncthreads();
  if (!is_last) return;
  if ((int)threadIdx.x < sg.nseg) {
    unsigned long long* const st = sg.state[threadIdx.x];
    const unsigned long long fp = __hip_atomic_load(&st[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long old = __hip_atomic_load(&st[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool dirty = ((sg.force >> threadIdx.x) & 1ull) || fp != old;
    __hip_atomic_store(&st[1], fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&st[2], dirty ? 1ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&st[3], st[3] + (dirty ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // rebuild count (diagnostic, tests)
    __hip_atomic_store(&st[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // accumulator left zeroed for the next call
  }
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// packed[(co*T + t)*C4 + c] <- natural[co][c][t] for the segments marked dirty (the layout of bt_pack_params). A work item is (row co,
// chunk of 64 channels): its 64 x T natural floats are one contiguous run -- read coalesced, transposed through LDS, written as T runs
// of 64 packed floats (a training step rebuilds every pack: read straight in packed order the natural tensors were fetched at a
// stride of T floats, 99 us per ResNet18 step).
constexpr int kPackCh = 64;
__global__ __launch_bounds__(256) void pack_dirty_kernel(PackSegs sg) {
  extern __shared__ float2 tile[];   // [T][64 + 1] (mu, sigma)
  int seg = 0;
  while (seg + 1 < sg.nseg && (int)blockIdx.x >= sg.first_block[seg + 1]) ++seg;
  if (__hip_atomic_load(&sg.state[seg][2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) return;
  const int nb = sg.first_block[seg + 1] - sg.first_block[seg], lb = blockIdx.x - sg.first_block[seg];
  const long long C = sg.C[seg], T = sg.T[seg], C4 = (C + 3) & ~3ll;
  const long long Co = sg.np[seg] / (T * C4);
  const long long cchunks = (C4 + kPackCh - 1) / kPackCh, items = Co * cchunks;
  const float* __restrict__ mu = sg.src_mu[seg];
  const float* __restrict__ rho = sg.src_rho[seg];
  float* __restrict__ mu_p = sg.mu_p[seg];
  float* __restrict__ sg_p = sg.sg_p[seg];
  const int Ti = (int)T;
  for (long long it = lb; it < items; it += nb) {
    const long long co = it / cchunks, c0 = (it - co * cchunks) * kPackCh;
    const int nc = (int)((C - c0) < kPackCh ? (C - c0 > 0 ? C - c0 : 0) : kPackCh);      // real channels of this chunk
    const int nc4 = (int)((C4 - c0) < kPackCh ? (C4 - c0) : kPackCh);                     // packed channels (padding holds zeros)
    const long long src0 = (co * C + c0) * T;
    __syncthreads();   // (the previous item's tile has been written out)
    for (int i = threadIdx.x; i < nc * Ti; i += 256) {
      const int c = i / Ti, t = i - c * Ti;
      tile[t * (kPackCh + 1) + c] = make_float2(mu[src0 + i], softplus(rho[src0 + i]));
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Ti * nc4; j += 256) {
      const int t = j / nc4, c = j - t * nc4;
      const float2 v = c < nc ? tile[t * (kPackCh + 1) + c] : make_float2(0.f, 0.f);
      const long long dst = (co * T + t) * C4 + c0 + c;
      mu_p[dst] = v.x;
      sg_p[dst] = v.y;
    }
  }
}

}  // namespace bt

extern "C" int bt_pack_sync(int32_t n_segments, const bt_pack_seg* segs, void* workspace, size_t workspace_bytes, bt_stream_t stream) {
  using namespace bt;
  if (n_segments <= 0 || n_segments > BT_PACK_MAX_SEGMENTS) return set_error(BT_ERR_BAD_ARG, "bt_pack_sync: n_segments must be in [1, 64]");
  if (!segs) return set_error(BT_ERR_BAD_ARG, "bt_pack_sync: null argument");
  if (!workspace || workspace_bytes < BT_WORKSPACE_BYTES) return set_error(BT_ERR_WORKSPACE, "bt_pack_sync: workspace smaller than BT_WORKSPACE_BYTES");
  PackSegs fp, pk;
  int fblocks = 0, pblocks = 0;
  fp.force = 0ull;
  for (int i = 0; i < n_segments; ++i) {
    const bt_pack_seg& s = segs[i];
    if (!s.mu_w || !s.rho_w || !s.mu_packed || !s.sigma_packed || !s.state || s.Co <= 0 || s.Ci <= 0 || s.taps <= 0)
      return set_error(BT_ERR_BAD_ARG, "bt_pack_sync: null pointer or non-positive dimension in a segment");
    if ((s.src_mu == nullptr) != (s.src_rho == nullptr)) return set_error(BT_ERR_BAD_ARG, "bt_pack_sync: src_mu and src_rho must both be given or both be NULL");
    fp.mu[i] = s.mu_w, fp.rho[i] = s.rho_w;
    fp.src_mu[i] = s.src_mu ? s.src_mu : s.mu_w, fp.src_rho[i] = s.src_rho ? s.src_rho : s.rho_w;
    fp.mu_p[i] = s.mu_packed, fp.sg_p[i] = s.sigma_packed;
    fp.state[i] = reinterpret_cast<unsigned long long*>(s.state);
    fp.n[i] = s.Co * s.Ci * s.taps;
    fp.C[i] = s.Ci, fp.T[i] = s.taps, fp.np[i] = s.Co * s.taps * ((s.Ci + 3) & ~3ll);
    if (s.force) fp.force |= 1ull << i;
  }
  pk = fp;
  long long max_taps = 1;
  for (int i = 0; i < n_segments; ++i) {
    const bt_pack_seg& s_ = segs[i];
    long long nb = (fp.n[i] + kFpElemsPerBlock - 1) / kFpElemsPerBlock;
    if (nb > 128) nb = 128;   // (more blocks are slower: every wave ends in one 64-bit atomic on the segment's accumulator -- 1024 blocks: 14 -> 25 us for 1.5 M weights)
    fp.first_block[i] = fblocks, fblocks += (int)nb;
    const long long C4 = (s_.Ci + 3) & ~3ll;
    long long pb = s_.Co * ((C4 + kPackCh - 1) / kPackCh);   // (row, 64-channel chunk) work items
    if (pb > 256) pb = 256;
    pk.first_block[i] = pblocks, pblocks += (int)pb;
    if (s_.taps > max_taps) max_taps = s_.taps;
  }
  fp.first_block[n_segments] = fblocks, pk.first_block[n_segments] = pblocks;
  fp.nseg = pk.nseg = n_segments;
  hipLaunchKernelGGL(pack_fingerprint_kernel, dim3(fblocks), dim3(kFpThreads), 0, (hipStream_t)stream, fp, ws_counter(workspace), fblocks);
  if (int rc = check_launch("bt_pack_sync (fingerprint)")) return rc;
  const size_t lds = (size_t)max_taps * (kPackCh + 1) * sizeof(float2);
  if (max_taps > 128) return set_error(BT_ERR_UNSUPPORTED, "bt_pack_sync: kernels larger than 128 taps are not supported");
  if (lds > 48 * 1024) {   // (above the default dynamic-LDS limit: 11 x 11 kernels and larger)
    static bool flags[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "bt_pack_sync: hipGetDevice failed");
    if (!flags[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(pack_dirty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * (kPackCh + 1) * (int)sizeof(float2)) != hipSuccess)
        return set_error(BT_ERR_HIP_BASE, "bt_pack_sync: cannot raise the dynamic LDS limit");
      flags[dev] = true;
    }
  }
  hipLaunchKernelGGL(pack_dirty_kernel, dim3(pblocks), dim3(256), lds, (hipStream_t)stream, pk);
  return check_launch("bt_pack_sync (pack)");
}
