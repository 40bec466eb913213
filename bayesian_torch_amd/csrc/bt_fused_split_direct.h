// "Direct" flavour of the split-precision fused forward for 1x1 / stride-1 convolutions with at most 256 input channels per group
// (the bottleneck ResNets' expanding and reducing 1x1 layers: K = 64 ... 256, M = B*H*W in the hundreds of thousands).
//
// The general split kernel (bt_fused_split.h) gives such a layer one-step K-stages -- a 512-pixel x tile leaves room for two
// channel octets per stage -- behind a workgroup barrier and a one-stage-deep load, and a fresh prologue, weight draw and output
// stage per 512 pixels: a 64 x 512 x 256 tile is 12.8 MFLOP, and its workgroup lives ~140 K cycles for ~25 K cycles of MFMA
// (ResNet50 / ImageNet layer3 conv3: 48 TFLOP/s fp32-equivalent). Here instead:
//   * the SAMPLED weight tile of a (group, 64-channel tile, MC sample) -- 64 rows x K, as bf16 pieces: 6 KB per K16 step, 96 KB at
//     K = 256 -- is synthesised ONCE per workgroup (same Philox blocks, same arithmetic as everywhere else) and stays in LDS;
//   * a workgroup is persistent: its 8 waves each walk 64-pixel sub-tiles of the (group, tile, sample)'s pixel range, with NO
//     barrier in the steady state -- while one wave of a SIMD stores its outputs the other one multiplies;
//   * a 1x1 convolution's x operand has no reuse across waves (a wave owns all 64 output channels of its pixels) and no window
//     overlap, so it does not go through LDS at all: a lane fetches the 8 channels of its pixel for a K16 half-step straight into
//     registers (consecutive lanes = consecutive pixels = 128-byte runs per channel), two K32 groups ahead of their use across
//     sub-tile boundaries, splits them into the three bf16 pieces in registers and feeds the MFMA's B operand;
//   * the MFMA runs with the weights as A and the pixels as B, so an accumulator register is ONE channel x 32 consecutive pixels:
//     the output stage is plain coalesced 128-byte stores (and residual loads) from registers -- no LDS staging, no barrier.
// K > 256 (RESIDENT = false; the reducing 1x1 layers and the deep downsamples: K = 512 ... 2048): the weight image does not fit, so
// it is streamed through two LDS buffers in chunks of 8 K16 steps (48 KB each) that ALL 8 waves synthesise between their own MFMA
// steps -- one workgroup barrier per chunk (192 MFMAs per wave) instead of one per step -- and re-drawn per 512 pixels like the
// general kernel does; x, the persistent walk and the output stage are the same.
// Strided 1x1 convolutions (the downsamples) only change the pixel -> input position map of the x fetch.
// Canonical K order, term order, draw stream and output-stage arithmetic are those of bt_fused_split.h (one active tap: consecutive
// octets in pairs), so the results are bit-identical to the general kernel's (tests/test_gpu_round3.py).
#pragma once
#include "bt_fused_split.h"

namespace bt {

constexpr int kDirectMaxK = 256;
constexpr int kDirectThreads = 512;
constexpr int kDirectWStep = 2 * 3 * 64 * 16;  // bytes of one K16 step of the weight image: [lane half][piece][row][8 ch] bf16
constexpr int kDirectChunk = 8;                // K16 steps per streamed weight chunk
inline int direct_lds_bytes(int Cig) { return (Cig <= kDirectMaxK ? (Cig >> 4) : 2 * kDirectChunk) * kDirectWStep + 64 * 16 + 64; }

template <bool RESIDENT>
__global__ __launch_bounds__(kDirectThreads) void fused_split_direct_kernel(const FwdArgs a) {
  constexpr int BN = 64, NP = 3, W_STEP = kDirectWStep, W_HALF = NP * BN * 16, W_PIECE = BN * 16;
  constexpr int TN = 2, TM = 2;  // a wave: 64 channels x 64 pixels
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int G8 = a.Cig >> 3, nsteps = a.Cig >> 4;  // host: Cig % 64 == 0 -- the K loop runs in blocks of four K16 steps
  char* const wbuf = smem_c;
  float4* const cst = reinterpret_cast<float4*>(smem_c + (RESIDENT ? nsteps : 2 * kDirectChunk) * W_STEP);  // per channel of the tile: (bias, scale, shift, -)

  // ---- workgroup -> (group, sample, chunk of the pixel range, channel tile); the channel tile runs fastest, so the workgroups that
  // read the same pixels are neighbours on one XCD and share its L2
  int L = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, a.total_blocks));
  int Lq = udiv_inv(L, a.n_tiles, a.inv_n_tiles);
  const int nt = __builtin_amdgcn_readfirstlane(L - Lq * a.n_tiles);
  L = Lq, Lq = udiv_inv(L, a.m_tiles, a.inv_m_tiles);
  const int chunk = __builtin_amdgcn_readfirstlane(L - Lq * a.m_tiles);
  L = Lq, Lq = udiv_inv(L, a.S, a.inv_S);
  const int s = __builtin_amdgcn_readfirstlane(L - Lq * a.S);
  const int g = __builtin_amdgcn_readfirstlane(Lq);
  const int n0 = nt * BN;
  const uint32_t sample = a.sample0 + (uint32_t)s;
  const int Cig = a.Cig, HW = a.HW, HWo = a.HoWo;

  RngKey key_w;
  key_w.seed_lo = a.seed_lo;
  key_w.seed_hi = a.seed_hi;
  key_w.call = a.call + (a.call_base ? __builtin_nontemporal_load(a.call_base) : 0u);
  key_w.layer_tensor = layer_tensor_word(a.layer_id, 0);

  // ---- KL: the first kl_slices workgroups sweep a slice of the natural-layout parameters each (all 8 waves), before anything else
  const bool kl_block = a.do_kl && (int)blockIdx.x < a.kl_slices;
  if (kl_block) {
    long long chunk_e = (a.w_elems + a.kl_slices - 1) / a.kl_slices;
    chunk_e = (chunk_e + 3) & ~3ll;
    const long long lo = (long long)blockIdx.x * chunk_e;
    const long long hi = (lo + chunk_e < a.w_elems) ? lo + chunk_e : a.w_elems;
    const bool v4 = ((((uintptr_t)a.mu_w | (uintptr_t)a.rho_w | (uintptr_t)a.pmu_w | (uintptr_t)a.psig_w) & 15u) == 0);
    double acc = 0.0;
    long long i = lo + 4ll * tid;
    if (v4) {
      // four groups per trip, all 16 loads in flight before the first use (ResNet18 / CIFAR layer4: 18 groups per thread -- one at a time
      // that is 18 exposed memory round trips, ~45 K cycles, at the head of a 100 K-cycle workgroup); same per-thread order of accumulation
      while (i + 3 < hi) {
        float4 m4[4], r4[4], p4[4], q4[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long iu = i + (long long)u * 4 * kDirectThreads;
          ok[u] = iu + 3 < hi;
          if (ok[u]) {
            m4[u] = *reinterpret_cast<const float4*>(a.mu_w + iu), r4[u] = *reinterpret_cast<const float4*>(a.rho_w + iu);
            p4[u] = *reinterpret_cast<const float4*>(a.pmu_w + iu), q4[u] = *reinterpret_cast<const float4*>(a.psig_w + iu);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (ok[u]) {
            const float t0 = kl_term(m4[u].x, softplus(r4[u].x), p4[u].x, q4[u].x) + kl_term(m4[u].y, softplus(r4[u].y), p4[u].y, q4[u].y);
            const float t1 = kl_term(m4[u].z, softplus(r4[u].z), p4[u].z, q4[u].z) + kl_term(m4[u].w, softplus(r4[u].w), p4[u].w, q4[u].w);
            acc += (double)t0 + (double)t1;
            i += 4 * kDirectThreads;
          }
        }
      }
    }
    for (; i < hi; i += 4 * kDirectThreads)  // tail quad / unaligned bases
      for (int j = 0; j < 4; ++j)
        if (i + j < hi) acc += (double)kl_term(a.mu_w[i + j], softplus(a.rho_w[i + j]), a.pmu_w[i + j], a.psig_w[i + j]);
    const double wsum = wave_sum(acc);
    // write-through partial per wave, drained; ONE ticket per workgroup behind the barrier below (bt_fused_split.h, kl_finish / kl_ticket)
    if (lane == 0) __hip_atomic_store(&a.slots[(int)blockIdx.x * 8 + wave], wsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // ---- sampled weights: unit u = (channel quad cq, row n, octet o) = one Philox block = 4 weights. synth_pair(dst, oct0, it) draws
  // units (2 it) * 512 + tid and (2 it + 1) * 512 + tid of the octets [oct0, ...) into the weight image at dst (their loads and
  // Philox chains interleave). RESIDENT: the whole tile once, here; else chunk by chunk between the MFMA steps below.
  const int pk_bytes = a.Co * a.T * Cig * 4;
  const __amdgpu_buffer_rsrc_t r_mu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mu_pk), 0, pk_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sig_pk), 0, pk_bytes, 0x00020000);
  auto synth_one = [&](char* dst_img, int oct0, int it, int noct) {   // unit it * 512 + tid of the octets [oct0, oct0 + noct)
    const int u = it * kDirectThreads + tid;
    const int cq = u & 1, n = (u >> 1) & (BN - 1), ol = u >> 7;
    const bool in = ol < noct;
    const bool rv = in && n0 + n < a.Cog;
    const uint32_t co = (uint32_t)(g * a.Cog + n0 + n);
    const uint32_t eo = (co * (uint32_t)a.T + (uint32_t)a.d_tap) * (uint32_t)Cig + (uint32_t)(8 * (oct0 + ol) + 4 * cq);   // the general kernels' draw index: tap-major
    const uint32_t sb = rv ? 4u * eo : 0x80000000u;  // rows past the tile's channels load zeros: w = 0 + 0 * eps
    const int st = ol >> 1, hf = ol & 1;
    char* const dst = dst_img + st * W_STEP + hf * W_HALF + (n ^ ((2 * st + hf) & 7)) * 16 + cq * 8;
    const float4 mu = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_mu, (int)sb, 0, 0));
    const float4 rs = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, (int)sb, 0, 0));
    float ep[4];
    philox_normal4(key_w, sample, eo >> 2, ep);
    const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, s4[4] = {rs.x, rs.y, rs.z, rs.w};
    uint32_t wh[4], wm_[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pieces(__fadd_rn(m4[j], __fmul_rn(s4[j], ep[j])), wh[j], wm_[j], wl[j]);
    if (in) {
      *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(wh[1], wh[0]), pack_hi16(wh[3], wh[2]));
      *reinterpret_cast<uint2*>(dst + W_PIECE) = make_uint2(pack_hi16(wm_[1], wm_[0]), pack_hi16(wm_[3], wm_[2]));
      *reinterpret_cast<uint2*>(dst + 2 * W_PIECE) = make_uint2(pack_hi16(wl[1], wl[0]), pack_hi16(wl[3], wl[2]));
    }
  };
  // The same unit in two halves for the streamed flavour: a unit's two 16-byte parameter loads are issued two K16 steps before it is
  // drawn, so that they are OLDER than the 32 x loads issued in between and the wait for them (vmcnt counts in order) does not drain
  // the x ring. All of it stays inside one chunk: nothing is carried around the loop.
  struct Unit {
    float4 mu, rs;
    uint32_t eo;
    int lds;  // byte offset in the weight image
  };
  auto unit_issue = [&](int oct0, int it) -> Unit {   // unit it * 512 + tid of a full chunk
    Unit t;
    const int u = it * kDirectThreads + tid;
    const int cq = u & 1, n = (u >> 1) & (BN - 1), ol = u >> 7;
    const bool rv = n0 + n < a.Cog;
    const uint32_t co = (uint32_t)(g * a.Cog + n0 + n);
    t.eo = (co * (uint32_t)a.T + (uint32_t)a.d_tap) * (uint32_t)Cig + (uint32_t)(8 * (oct0 + ol) + 4 * cq);
    const uint32_t sb = rv ? 4u * t.eo : 0x80000000u;
    const int st = ol >> 1, hf = ol & 1;
    t.lds = st * W_STEP + hf * W_HALF + (n ^ ((2 * st + hf) & 7)) * 16 + cq * 8;
    t.mu = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_mu, (int)sb, 0, 0));
    t.rs = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, (int)sb, 0, 0));
    return t;
  };
  auto unit_finish = [&](char* dst_img, const Unit& t) {
    float ep[4];
    philox_normal4(key_w, sample, t.eo >> 2, ep);
    const float m4[4] = {t.mu.x, t.mu.y, t.mu.z, t.mu.w}, s4[4] = {t.rs.x, t.rs.y, t.rs.z, t.rs.w};
    uint32_t wh[4], wm_[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pieces(__fadd_rn(m4[j], __fmul_rn(s4[j], ep[j])), wh[j], wm_[j], wl[j]);
    char* const dst = dst_img + t.lds;
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(wh[1], wh[0]), pack_hi16(wh[3], wh[2]));
    *reinterpret_cast<uint2*>(dst + W_PIECE) = make_uint2(pack_hi16(wm_[1], wm_[0]), pack_hi16(wm_[3], wm_[2]));
    *reinterpret_cast<uint2*>(dst + 2 * W_PIECE) = make_uint2(pack_hi16(wl[1], wl[0]), pack_hi16(wl[3], wl[2]));
  };
  auto synth_pair = [&](char* dst_img, int oct0, int it, int noct) {
    synth_one(dst_img, oct0, 2 * it, noct);
    synth_one(dst_img, oct0, 2 * it + 1, noct);
  };
  if constexpr (RESIDENT) {
    const int npairs = (2 * BN * G8 + 2 * kDirectThreads - 1) / (2 * kDirectThreads);
    for (int it = 0; it < npairs; ++it) synth_pair(wbuf, 0, it, G8);
  } else {
    synth_pair(wbuf, 0, 0, 2 * kDirectChunk);   // chunk 0 (2 x 64 x 16 octets = 2048 units: two pairs per thread)
    synth_pair(wbuf, 0, 1, 2 * kDirectChunk);
  }
  // bias draw + output-stage constants of the tile's channels
  if (tid < BN) {
    float bv = 0.f;
    const int co_g = n0 + tid;
    if (a.mu_b && co_g < a.Cog) {
      const int co = g * a.Cog + co_g;
      RngKey kb = key_w;
      kb.layer_tensor = layer_tensor_word(a.layer_id, 1);
      float z[4];
      philox_normal4(kb, sample, (uint32_t)(co >> 2), z);
      const int sel = co & 3;
      const float e = sel == 0 ? z[0] : sel == 1 ? z[1] : sel == 2 ? z[2] : z[3];
      bv = __fadd_rn(a.mu_b[co], __fmul_rn(softplus(a.rho_b[co]), e));
    }
    const bool cv = a.ep_scale && co_g < a.Cog;
    const int cs = cv ? g * a.Cog + co_g : 0;
    const float sc = a.ep_scale ? a.ep_scale[cs] : 1.f, sh = a.ep_shift ? a.ep_shift[cs] : 0.f;
    cst[tid] = make_float4(bv, cv ? sc : 1.f, cv ? sh : 0.f, 0.f);
  }
  __syncthreads();  // weights and constants staged; every wave's KL partial published
  if (kl_block && wave == 0) {
    const int nslots = 8 * a.kl_slices;
    int last = 0;
    if (lane == 0) last = (__hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.kl_slices - 1u) ? 1 : 0;
    if (__builtin_amdgcn_readfirstlane(last)) {  // this workgroup arrived last: every slot is published
      double t = 0.0;
      for (int q = lane; q < nslots; q += 64) t += __hip_atomic_load(&a.slots[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t = wave_sum(t);
      double bt_ = 0.0;
      if (a.mu_b)
        for (int c = lane; c < a.Co; c += 64) bt_ += (double)kl_term(a.mu_b[c], softplus(a.rho_b[c]), a.pmu_b[c], a.psig_b[c]);
      bt_ = wave_sum(bt_);
      if (lane == 0) {
        float kl = (float)(t / (double)a.w_elems);
        if (a.mu_b) kl += (float)(bt_ / (double)a.Co);
        a.kl_out[0] = kl;
        __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // leave the workspace zeroed
      }
    }
  }

  // ---- steady state: every wave for itself -----------------------------------------------------------------------------------------
  const float* const xs = a.x + (long long)s * a.x_sample_stride;
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, (int)(a.x_elems * 4), 0x00020000);
  float* const out_s = a.out + (long long)s * a.out_elems;
  const float* const res_s = a.ep_res ? a.ep_res + (long long)s * a.ep_res_stride : nullptr;
  const bool relu = a.ep_relu != 0;
  const int HWb = 4 * HW;
  const int nsub = (a.M + 63) >> 6;
  const int sub_lo = chunk * a.t_NI;                                       // t_NI: sub-tiles per chunk (host; RESIDENT = false: a multiple of 8)
  const int sub_hi = sub_lo + a.t_NI < nsub ? sub_lo + a.t_NI : nsub;
  const uint32_t inv_hwo = HWo > 1 ? (a.inv_rw ? a.inv_rw : inv32(HWo)) : 0u;  // inv_rw / inv_wt: ceil(2^32 / HoWo), ceil(2^32 / Wo) from the host
  const bool strided = a.SH != 1 || a.SW != 1;
  const uint32_t inv_wo = (strided && a.Wo > 1) ? (a.inv_wt ? a.inv_wt : inv32(a.Wo)) : 0u;
  auto px_decode = [&](int m, int& b, int& p, int& ipos) {  // pixel of the sample -> image, output position, input position
    b = HWo == 1 ? m : (int)__umulhi((uint32_t)m, inv_hwo);
    p = m - b * HWo;
    ipos = p;   // (a padded window over a 1x1 image: p = 0 and its one live tap reads input position 0)
    if (strided) {
      const int ho = a.Wo == 1 ? p : (int)__umulhi((uint32_t)p, inv_wo), wo = p - ho * a.Wo;
      ipos = ho * a.SH * a.W + wo * a.SW;
    }
  };

  // load head: position (sub-tile hs, K16 step hq) of the x stream, THREE steps ahead of the multiplier (a ring of four step
  // buffers, 16 registers each; the K loop is unrolled by four so the ring needs no copies -- host: Cig % 64 == 0)
  int hs = sub_lo + wave, hq = 0;
  uint32_t xo[TM];  // byte offset of (this lane's pixel, channel 8 * lh of the group g) or out of range
  auto head_setup = [&]() {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = hs * 64 + j * 32 + li;
      int b, p, ipos;
      px_decode(m, b, p, ipos);
      const bool ok = hs < sub_hi && m < a.M;
      xo[j] = ok ? (uint32_t)(((b * a.Ci + g * Cig + 8 * lh) * HW + ipos) * 4) : 0x80000000u;
    }
  };
  auto head_load = [&](float (&d)[TM][8]) {
    const int cb = (16 * hq) * HWb;  // uniform: first channel of the step
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) d[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_x, (int)xo[j], cb + c * HWb, 0));
    if (++hq == nsteps) {
      hq = 0, hs += 8;
      head_setup();
    }
  };
  float xr[4][TM][8];
  head_setup();
  head_load(xr[0]);
  head_load(xr[1]);
  head_load(xr[2]);

  // Four K16 steps from the weight image at wq (its step 0 = the block's first step), x from the ring.
  // (Round 3, measured with A/B builds on one box, tools/build_variant.sh + tools/ab_direct.sh, S = 8, b256: emitting the split of item
  // t + 1 ahead of item t's twelve MFMAs, alone or spread between them by sched_group_barrier -- one MFMA, then 4 or 8 VALU -- is 1 ... 4 %
  // SLOWER than this plain form on ResNet50's layer3 conv3 / conv1, layer1 conv3 and the layer2 downsample (1443 / 1895 / 2241 / 3808 us
  // here; 1485 ... 1512 / 1881 ... 1953 / 2288 ... 2378 / 3742 ... 3804 us pipelined). PMC of the conv3 launch: matrix pipe busy 52 %,
  // VALU busy 45 % of a shader clock that averages ~1.35 GHz of the 2.4 GHz it could run at: the launch sits at the power limit, and
  // what it needs is fewer instructions and bytes per MFMA, not a denser schedule.)
  f32x16 acc[TN][TM];
  auto four_steps = [&](const char* wq, auto&& mid) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u == 2) mid();
      head_load(xr[(u + 3) & 3]);
      const char* const wp = wq + u * W_STEP + lh * W_HALF + (li ^ ((2 * u + lh) & 7)) * 16;   // (the swizzle of step Q is (2 Q + lh) & 7 = (2 u + lh) & 7: blocks start at multiples of 4)
      bf16x8 wf[TN][NP];
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) wf[i][p] = *reinterpret_cast<const bf16x8*>(wp + p * W_PIECE + i * 32 * 16);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        uint32_t hq[4], mq[4], lq[4];
        if constexpr (RESIDENT) {
          uint32_t ph[8], pm[8], pl[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) split_pieces(xr[u][j][c], ph[c], pm[c], pl[c]);
#pragma unroll
          for (int c = 0; c < 4; ++c) hq[c] = pack_hi16(ph[2 * c + 1], ph[2 * c]), mq[c] = pack_hi16(pm[2 * c + 1], pm[2 * c]), lq[c] = pack_hi16(pl[2 * c + 1], pl[2 * c]);
        } else {
          // (the packed-fp32 form of the same split: measured on cfg5, -4..-12 % on the streamed layers, +3..+10 % on the resident ones)
#pragma unroll
          for (int c = 0; c < 4; ++c) split_pair(xr[u][j][2 * c], xr[u][j][2 * c + 1], hq[c], mq[c], lq[c]);
        }
        const uint4 h4 = make_uint4(hq[0], hq[1], hq[2], hq[3]), m4 = make_uint4(mq[0], mq[1], mq[2], mq[3]), l4 = make_uint4(lq[0], lq[1], lq[2], lq[3]);
        const bf16x8 x0 = __builtin_bit_cast(bf16x8, h4), x1 = __builtin_bit_cast(bf16x8, m4), x2 = __builtin_bit_cast(bf16x8, l4);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          // D[channel][pixel]: the weights are the A operand, the pixels the B operand; the six terms in the general kernel's order
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], x0, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][1], x0, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], x1, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][2], x0, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][1], x1, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], x2, acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  // RESIDENT: a wave walks its own sub-tiles (sub_lo + wave, + 8, ...), no barriers. Else the 8 waves walk 512-pixel tiles together
  // (a wave's sub-tile may lie past the end: its loads return zeros and nothing is stored) and meet at one barrier per weight chunk.
  const int n_iter = RESIDENT ? 0 : (sub_hi - sub_lo + 7) >> 3;
  const int NC = RESIDENT ? 1 : nsteps / kDirectChunk;
  int gc = 0;  // chunks consumed so far: buffer gc & 1 holds the current one
  for (int sub = sub_lo + wave, it_ = 0; RESIDENT ? sub < sub_hi : it_ < n_iter; sub += 8, ++it_) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (RESIDENT) {
      for (int q4 = 0; q4 < nsteps; q4 += 4) four_steps(wbuf + q4 * W_STEP, [] {});
    } else {
      for (int c = 0; c < NC; ++c, ++gc) {
        const char* const wcur = wbuf + (gc & 1) * (kDirectChunk * W_STEP);
        char* const wnext = wbuf + ((gc + 1) & 1) * (kDirectChunk * W_STEP);
        const int cn = c + 1 < NC ? c + 1 : 0;                 // the chunk after this one (the next tile starts over)
        // this thread's four units of the next chunk, one every two steps (their Philox chains run in the MFMAs' shadow)
        // this thread's four units of the next chunk, one every two steps: their Philox chains run in the MFMAs' shadow, their
        // parameter loads two steps ahead of them (see unit_issue)
        const int o_next = 2 * kDirectChunk * cn;
        Unit ua = unit_issue(o_next, 0), ub;
        four_steps(wcur, [&] { unit_finish(wnext, ua); ub = unit_issue(o_next, 1); });
        unit_finish(wnext, ub);
        ua = unit_issue(o_next, 2);
        four_steps(wcur + 4 * W_STEP, [&] { unit_finish(wnext, ua); ub = unit_issue(o_next, 3); });
        unit_finish(wnext, ub);
        __syncthreads();
      }
    }

    // ---- output stage: register r of accumulator (i, j) = channel i*32 + (r&3) + 8*(r>>2) + 4*lh, pixel j*32 + li
    asm volatile("" ::: "memory");  // (keeps the 32 float4 constants below in LDS: hoisted out of the sub-tile loop they would cost 128 registers -- and spill)
    uint32_t oo[TM];
    bool pv[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = sub * 64 + j * 32 + li;
      int b, p, ipos;
      px_decode(m, b, p, ipos);
      pv[j] = sub < sub_hi && m < a.M;
      oo[j] = pv[j] ? (uint32_t)((b * a.Co + g * a.Cog + n0) * HWo + p) : 0u;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      float rr[16][TM];
      if (res_s) {  // the residual values of this half of the channels in one batch, ahead of the first store
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
          for (int j = 0; j < TM; ++j) rr[r][j] = (pv[j] && n0 + c < a.Cog) ? res_s[oo[j] + (uint32_t)(c * HWo)] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float4 k = cst[c];
        const bool cok = n0 + c < a.Cog;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          float v = __fadd_rn(__fmul_rn(__fadd_rn(acc[i][j][r], k.x), k.y), k.z);
          if (res_s) v = __fadd_rn(v, rr[r][j]);
          v = (relu && v < 0.f) ? 0.f : v;
          if (pv[j] && cok) out_s[oo[j] + (uint32_t)(c * HWo)] = v;
        }
      }
    }
  }
}

}  // namespace bt
