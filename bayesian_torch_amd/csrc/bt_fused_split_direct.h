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
// Canonical K order, term order, draw stream and output-stage arithmetic are those of bt_fused_split.h (one active tap: consecutive
// octets in pairs), so the results are bit-identical to the general kernel's (tests/test_gpu_round3.py).
#pragma once
#include "bt_fused_split.h"

namespace bt {

constexpr int kDirectMaxK = 256;
constexpr int kDirectThreads = 512;
constexpr int kDirectWStep = 2 * 3 * 64 * 16;  // bytes of one K16 step of the weight image: [lane half][piece][row][8 ch] bf16
inline int direct_lds_bytes(int Cig) { return (Cig >> 4) * kDirectWStep + 64 * 16 + 64; }

__global__ __launch_bounds__(kDirectThreads) void fused_split_direct_kernel(const FwdArgs a) {
  constexpr int BN = 64, NP = 3, W_STEP = kDirectWStep, W_HALF = NP * BN * 16, W_PIECE = BN * 16;
  constexpr int TN = 2, TM = 2;  // a wave: 64 channels x 64 pixels
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int G8 = a.Cig >> 3, nsteps = a.Cig >> 4;  // host: Cig % 64 == 0 -- the K loop runs in blocks of four K16 steps
  char* const wbuf = smem_c;
  float4* const cst = reinterpret_cast<float4*>(smem_c + nsteps * W_STEP);  // per channel of the tile: (bias, scale, shift, -)

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
  const int Cig = a.Cig, HW = a.HW;

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
      for (; i + 3 < hi; i += 4 * kDirectThreads) {
        const float4 m4 = *reinterpret_cast<const float4*>(a.mu_w + i), r4 = *reinterpret_cast<const float4*>(a.rho_w + i);
        const float4 p4 = *reinterpret_cast<const float4*>(a.pmu_w + i), q4 = *reinterpret_cast<const float4*>(a.psig_w + i);
        const float t0 = kl_term(m4.x, softplus(r4.x), p4.x, q4.x) + kl_term(m4.y, softplus(r4.y), p4.y, q4.y);
        const float t1 = kl_term(m4.z, softplus(r4.z), p4.z, q4.z) + kl_term(m4.w, softplus(r4.w), p4.w, q4.w);
        acc += (double)t0 + (double)t1;
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

  // ---- the sampled weight tile, once: unit u = (channel quad cq, row n, octet o) = one Philox block = 4 weights
  {
    const int pk_bytes = a.Co * Cig * 4;
    const __amdgpu_buffer_rsrc_t r_mu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mu_pk), 0, pk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sig_pk), 0, pk_bytes, 0x00020000);
    const int nunits = 2 * BN * G8;
    for (int u0 = 0; u0 < nunits; u0 += 2 * kDirectThreads) {  // two units per thread and trip: their loads and Philox chains interleave
      float4 mu[2], rs[2];
      uint32_t eo[2];
      int lo_[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int u = u0 + k * kDirectThreads + tid;
        const int cq = u & 1, n = (u >> 1) & (BN - 1), o = u >> 7;
        const bool rv = u < nunits && n0 + n < a.Cog;
        const uint32_t co = (uint32_t)(g * a.Cog + n0 + n);
        eo[k] = co * (uint32_t)Cig + (uint32_t)(8 * o + 4 * cq);
        const uint32_t sb = rv ? 4u * eo[k] : 0x80000000u;  // rows past the tile's channels load zeros: w = 0 + 0 * eps
        mu[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_mu, (int)sb, 0, 0));
        rs[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, (int)sb, 0, 0));
        const int st = o >> 1, hf = o & 1;
        lo_[k] = u < nunits ? st * W_STEP + hf * W_HALF + (n ^ ((2 * st + hf) & 7)) * 16 + cq * 8 : -1;
      }
      float ep[2][4];
#pragma unroll
      for (int k = 0; k < 2; ++k) philox_normal4(key_w, sample, eo[k] >> 2, ep[k]);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float m4[4] = {mu[k].x, mu[k].y, mu[k].z, mu[k].w}, s4[4] = {rs[k].x, rs[k].y, rs[k].z, rs[k].w};
        uint32_t wh[4], wm_[4], wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split_pieces(__fadd_rn(m4[j], __fmul_rn(s4[j], ep[k][j])), wh[j], wm_[j], wl[j]);
        if (lo_[k] >= 0) {
          char* const dst = wbuf + lo_[k];
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(wh[1], wh[0]), pack_hi16(wh[3], wh[2]));
          *reinterpret_cast<uint2*>(dst + W_PIECE) = make_uint2(pack_hi16(wm_[1], wm_[0]), pack_hi16(wm_[3], wm_[2]));
          *reinterpret_cast<uint2*>(dst + 2 * W_PIECE) = make_uint2(pack_hi16(wl[1], wl[0]), pack_hi16(wl[3], wl[2]));
        }
      }
    }
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
  const int sub_lo = chunk * a.t_NI;                                       // t_NI: sub-tiles per chunk (host)
  const int sub_hi = sub_lo + a.t_NI < nsub ? sub_lo + a.t_NI : nsub;
  const uint32_t inv_hw = HW > 1 ? (a.inv_rw ? a.inv_rw : inv32(HW)) : 0u;  // inv_rw: ceil(2^32 / HW) from the host
  auto px_decode = [&](int m, int& b, int& hw) {
    b = HW == 1 ? m : (int)__umulhi((uint32_t)m, inv_hw);
    hw = m - b * HW;
  };

  // load head: position (sub-tile hs, K16 step hq) of the x stream, THREE steps ahead of the multiplier (a ring of four step
  // buffers, 16 registers each; the K loop is unrolled by four so the ring needs no copies -- host: Cig % 64 == 0)
  int hs = sub_lo + wave, hq = 0;
  uint32_t xo[TM];  // byte offset of (this lane's pixel, channel 8 * lh of the group g) or out of range
  auto head_setup = [&]() {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = hs * 64 + j * 32 + li;
      int b, hw;
      px_decode(m, b, hw);
      const bool ok = hs < sub_hi && m < a.M;
      xo[j] = ok ? (uint32_t)(((b * a.Ci + g * Cig + 8 * lh) * HW + hw) * 4) : 0x80000000u;
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

  for (int sub = sub_lo + wave; sub < sub_hi; sub += 8) {
    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int q4 = 0; q4 < nsteps; q4 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        head_load(xr[(u + 3) & 3]);
        const int Q = q4 + u;
        const char* const wp = wbuf + Q * W_STEP + lh * W_HALF + (li ^ ((2 * u + lh) & 7)) * 16;   // ((2 Q + lh) & 7 == (2 u + lh) & 7: q4 % 4 == 0)
        bf16x8 wf[TN][NP];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int p = 0; p < NP; ++p) wf[i][p] = *reinterpret_cast<const bf16x8*>(wp + p * W_PIECE + i * 32 * 16);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          uint32_t ph[8], pm[8], pl[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) split_pieces(xr[u][j][c], ph[c], pm[c], pl[c]);
          const uint4 h4 = make_uint4(pack_hi16(ph[1], ph[0]), pack_hi16(ph[3], ph[2]), pack_hi16(ph[5], ph[4]), pack_hi16(ph[7], ph[6]));
          const uint4 m4 = make_uint4(pack_hi16(pm[1], pm[0]), pack_hi16(pm[3], pm[2]), pack_hi16(pm[5], pm[4]), pack_hi16(pm[7], pm[6]));
          const uint4 l4 = make_uint4(pack_hi16(pl[1], pl[0]), pack_hi16(pl[3], pl[2]), pack_hi16(pl[5], pl[4]), pack_hi16(pl[7], pl[6]));
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
    }

    // ---- output stage: register r of accumulator (i, j) = channel i*32 + (r&3) + 8*(r>>2) + 4*lh, pixel j*32 + li
    asm volatile("" ::: "memory");  // (keeps the 32 float4 constants below in LDS: hoisted out of the sub-tile loop they would cost 128 registers -- and spill)
    uint32_t oo[TM];
    bool pv[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = sub * 64 + j * 32 + li;
      int b, hw;
      px_decode(m, b, hw);
      pv[j] = m < a.M;
      oo[j] = pv[j] ? (uint32_t)((b * a.Co + g * a.Cog + n0) * HW + hw) : 0u;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      float rr[16][TM];
      if (res_s) {  // the residual values of this half of the channels in one batch, ahead of the first store
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
          for (int j = 0; j < TM; ++j) rr[r][j] = (pv[j] && n0 + c < a.Cog) ? res_s[oo[j] + (uint32_t)(c * HW)] : 0.f;
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
          if (pv[j] && cok) out_s[oo[j] + (uint32_t)(c * HW)] = v;
        }
      }
    }
  }
}

}  // namespace bt
