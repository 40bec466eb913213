// "Skinny" flavour of the split-precision fused forward: NARROW HEADS whose output map is one pixel -- the classifier head (Linear
// 512 -> 10), small Linear layers, narrow convolutions that end on 1x1 maps.
//
// Such a layer is a tall-and-skinny GEMM per MC sample: M = B columns, K = (live taps) x Ci, ONE or two 64-row channel tiles. The general
// kernel gives it one workgroup per (sample, channel tile) -- 32 workgroups on a 256-CU part for ResNet18's head at 32 samples -- each
// walking K / 16 barrier stages behind a ~18 K-cycle prologue. Here:
//   * SPLIT-K: a workgroup = (channel tile, sample, 128-column tile, K-slice of 64 / 128 channels of ONE live tap): a 6-KB-per-step
//     weight image of 4 or 8 steps that it draws once, 4 waves of 64 channels x 32 columns, x straight into registers, no stage loop,
//     three barriers.
//   * The slices' fp32 partial tiles go to a scratch slab each; the LAST slice to arrive (one agent-scope ticket per tile) adds
//     them IN SLICE ORDER -- deterministic, whatever order they ran in -- and applies the output stage (bias draw, folded BatchNorm,
//     residual, ReLU). The hand-off is cdna_hip_programming.md's write-through form (Guideline 16, R1): 16-byte sc1 slab stores,
//     every wave's vmcnt(0), barrier, ONE relaxed agent-scope ticket add; the reducer reads the slabs with sc1 loads. No fence
//     anywhere. Tickets live in the zeroed workspace and are reset by the reducer.
// Draw stream, piece arithmetic and output-stage op order are the other kernels'; the K order is this flavour's own (slices in index
// order: tap-major, then channels; consecutive octets in pairs inside a slice) and depends on the layer's geometry alone.
// WHERE IT PAYS (launch_skinny's gate, bt_fused_split.hip: at most 8 workgroups per sample): the head 37.6 -> 23.8 us. It was built for
// all of ResNet18 / CIFAR layer4 and does NOT pay there: with 16 ... 64 slices per sample the slabs' HBM round trip (64 KB per
// workgroup), the parameter / x / KL traffic of 4 workgroups per CU through one 64 B/clk L1 path and a second round of workgroups
// (1024 ... 2048 for 768 slots) cost what the shorter chains save (60 vs 53 ... 60 us; 140 vs 71 with four live taps). DESIGN.md 4.0b.
#pragma once
#include "bt_fused_split.h"

namespace bt {

constexpr int kSkinnyThreads = 256;
constexpr int kSkinnyCols = 128;                    // columns (images) of a tile: 4 waves x 32
constexpr int kSkinnyTicketSlot = 4000;             // tickets: the upper half of the workspace's slot array, as uint32 [8000]
constexpr int kSkinnyMaxTiles = 8000;
inline int skinny_lds_bytes(int ks) { return (ks >> 4) * kDirectWStep + 64 * 16 + 64; }

__global__ __launch_bounds__(kSkinnyThreads, 3) void fused_split_skinny_kernel(const FwdArgs a) {
  constexpr int BN = 64, NP = 3, W_STEP = kDirectWStep, W_HALF = NP * BN * 16, W_PIECE = BN * 16, TN = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int KS = a.sk_ks, nsteps = KS >> 4, NSL = a.sk_nsl, CPT = a.sk_cpt;   // slice width (64 | 128 channels), slices per tile, per tap
  char* const wbuf = smem_c;
  float4* const cst = reinterpret_cast<float4*>(smem_c + nsteps * W_STEP);
  int* const flag = reinterpret_cast<int*>(cst + 64);
  double* const klp = reinterpret_cast<double*>(flag + 4);

  // workgroup -> (group, sample, column tile, channel tile, slice); the slices of a tile are neighbours (one XCD after the remap)
  int L = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, a.total_blocks));
  int Lq = L / NSL;
  const int q = __builtin_amdgcn_readfirstlane(L - Lq * NSL);
  const int tile = Lq;                         // ticket / slab index
  L = Lq, Lq = L / a.n_tiles;
  const int nt = __builtin_amdgcn_readfirstlane(L - Lq * a.n_tiles);
  L = Lq, Lq = L / a.m_tiles;
  const int mt = __builtin_amdgcn_readfirstlane(L - Lq * a.m_tiles);
  L = Lq, Lq = L / a.S;
  const int s = __builtin_amdgcn_readfirstlane(L - Lq * a.S);
  const int g = __builtin_amdgcn_readfirstlane(Lq);
  const int n0 = nt * BN, Cig = a.Cig, T = a.T;
  const uint32_t sample = a.sample0 + (uint32_t)s;

  // the slice's tap: the (q / CPT)-th tap (row-major over the rectangle of live taps: the host's sk_kh0 / sk_nh / sk_kw0 / sk_nw) and
  // the one input pixel it reads
  const int want = q / CPT;
  const int ta = want / a.sk_nw, tb = want - ta * a.sk_nw;
  const int kh_ = a.sk_kh0 + ta, kw_ = a.sk_kw0 + tb;
  const int tap = kh_ * a.KW + kw_;
  const int pix = (kh_ * a.DH - a.PH) * a.W + (kw_ * a.DW - a.PW);
  const int cs0 = (q - want * CPT) * KS;   // first channel of the slice

  // ---- x addressing (column = image b, one input pixel per image)
  const float* const xs = a.x + (long long)s * a.x_sample_stride;
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, (int)(a.x_elems * 4), 0x00020000);
  const int b = mt * kSkinnyCols + wave * 32 + li;
  const bool bok = b < a.B;
  const uint32_t xo = bok ? (uint32_t)((((b * a.Ci + g * Cig + cs0 + 8 * lh) * a.HW) + pix) * 4) : 0x80000000u;
  const int HWb = 4 * a.HW;
  // ... and the parameters of this thread's weight units, and (threads 64..127) of the tile's bias / output-stage constants
  constexpr int UMAXS = 2 * BN * 16 / kSkinnyThreads;   // 8
  const int nunits = 2 * BN * (KS >> 3);
  float4 wmu[UMAXS], wrs[UMAXS];
  uint32_t weo[UMAXS];
  {
    const int pk_bytes = a.Co * T * Cig * 4;
    const __amdgpu_buffer_rsrc_t r_mu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mu_pk), 0, pk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sig_pk), 0, pk_bytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < UMAXS; ++k) {
      const int u = k * kSkinnyThreads + tid;
      const int cq = u & 1, n = (u >> 1) & (BN - 1), ol = u >> 7;
      const bool rv = u < nunits && n0 + n < a.Cog;
      const uint32_t co = (uint32_t)(g * a.Cog + n0 + n);
      weo[k] = (co * (uint32_t)T + (uint32_t)tap) * (uint32_t)Cig + (uint32_t)(cs0 + 8 * ol + 4 * cq);   // the draw index of every kernel: tap-major
      const uint32_t sb = rv ? 4u * weo[k] : 0x80000000u;   // rows past the tile's channels (and units past the slice) load zeros: w = 0 + 0 * eps
      wmu[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_mu, (int)sb, 0, 0));
      wrs[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, (int)sb, 0, 0));
    }
  }
  float c_mub = 0.f, c_rhob = 0.f, c_sc = 1.f, c_sh = 0.f;
  bool c_bias = false;
  if (tid >= 64 && tid < 128) {
    const int co_g = n0 + tid - 64;
    const bool in = co_g < a.Cog;
    const int co = g * a.Cog + (in ? co_g : 0);
    c_bias = a.mu_b && in;
    if (c_bias) c_mub = a.mu_b[co], c_rhob = a.rho_b[co];
    if (a.ep_scale && in) c_sc = a.ep_scale[co], c_sh = a.ep_shift[co];
  }

  RngKey key_w;
  key_w.seed_lo = a.seed_lo;
  key_w.seed_hi = a.seed_hi;
  key_w.call = a.call + (a.call_base ? __builtin_nontemporal_load(a.call_base) : 0u);
  key_w.layer_tensor = layer_tensor_word(a.layer_id, 0);

  // ---- KL: the first kl_slices workgroups (all of them up to 2048: a few elements per thread, so that no workgroup carries a long
  // sweep on top of its slice) sweep a slice of the natural-layout parameters each
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
      while (i + 3 < hi) {   // four groups per trip, all 16 loads in flight before the first use
        float4 m4[4], r4[4], p4[4], q4[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long iu = i + (long long)u * 4 * kSkinnyThreads;
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
            i += 4 * kSkinnyThreads;
          }
        }
      }
    }
    for (; i < hi; i += 4 * kSkinnyThreads)  // tail quad / unaligned bases
      for (int j = 0; j < 4; ++j)
        if (i + j < hi) acc += (double)kl_term(a.mu_w[i + j], softplus(a.rho_w[i + j]), a.pmu_w[i + j], a.psig_w[i + j]);
    const double wsum = wave_sum(acc);
    if (lane == 0) klp[wave] = wsum;   // the four waves' partials meet behind barrier 1
  }

  // ---- the slice's sampled weights: unit u = (channel quad cq, row n, octet ol) = one Philox block = 4 weights; 8 units per thread at
  // 128 channels. (Their parameter loads were issued at the top -- wmu / wrs -- together with x: a workgroup of this kernel is a chain
  // of memory round trips, ~2 us each on cold parameters, and every one taken off the chain counts: fetched trip by trip the draws
  // alone were 4 x (2 us wait + 1.5 us of Philox).)
#pragma unroll
  for (int k = 0; k < UMAXS; ++k) {
    if (k * kSkinnyThreads < nunits) {   // uniform
      float ep[4];
      philox_normal4(key_w, sample, weo[k] >> 2, ep);
      const float m4[4] = {wmu[k].x, wmu[k].y, wmu[k].z, wmu[k].w}, s4[4] = {wrs[k].x, wrs[k].y, wrs[k].z, wrs[k].w};
      uint32_t wh[4], wm_[4], wl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split_pieces(__fadd_rn(m4[j], __fmul_rn(s4[j], ep[j])), wh[j], wm_[j], wl[j]);
      const int u = k * kSkinnyThreads + tid;
      const int cq = u & 1, n = (u >> 1) & (BN - 1), ol = u >> 7, st = ol >> 1, hf = ol & 1;
      char* const dst = wbuf + st * W_STEP + hf * W_HALF + (n ^ ((2 * st + hf) & 7)) * 16 + cq * 8;
      *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(wh[1], wh[0]), pack_hi16(wh[3], wh[2]));
      *reinterpret_cast<uint2*>(dst + W_PIECE) = make_uint2(pack_hi16(wm_[1], wm_[0]), pack_hi16(wm_[3], wm_[2]));
      *reinterpret_cast<uint2*>(dst + 2 * W_PIECE) = make_uint2(pack_hi16(wl[1], wl[0]), pack_hi16(wl[3], wl[2]));
    }
  }
  // x: this lane's 8 channels of every K16 step. Issued HERE, behind the draws (its 64 registers do not overlap the parameters' then: the
  // kernel stays under 168 VGPRs, three workgroups per CU, and one workgroup's memory round trips hide behind the others' Philox chains)
  float xr[8][8];
#pragma unroll
  for (int Q = 0; Q < 8; ++Q) {
    if (Q < nsteps) {
      if (a.x_vec) {   // one pixel per image, aligned rows: the 8 channels are 32 contiguous bytes
        const float4 v0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)xo, 64 * Q, 0));
        const float4 v1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)xo, 64 * Q + 16, 0));
        xr[Q][0] = v0.x, xr[Q][1] = v0.y, xr[Q][2] = v0.z, xr[Q][3] = v0.w, xr[Q][4] = v1.x, xr[Q][5] = v1.y, xr[Q][6] = v1.z, xr[Q][7] = v1.w;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) xr[Q][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_x, (int)xo, (16 * Q + c) * HWb, 0));
      }
    }
  }

  __syncthreads();  // the weight image is staged; every wave's KL partial is published
  if (kl_block && wave == 0) {
    const int nslots = a.kl_slices;
    int last = 0;
    if (lane == 0) {   // one slot per workgroup (its waves' partials in wave order), stored to L2 and drained ahead of the ticket
      __hip_atomic_store(&a.slots[(int)blockIdx.x], ((klp[0] + klp[1]) + klp[2]) + klp[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last = (__hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.kl_slices - 1u) ? 1 : 0;
    }
    if (__builtin_amdgcn_readfirstlane(last)) {
      double t = 0.0;
      for (int k = lane; k < nslots; k += 512) {   // eight slots in flight per trip; added in slot order per lane
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = k + 64 * j < nslots ? __hip_atomic_load(&a.slots[k + 64 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += v[j];
      }
      t = wave_sum(t);
      double bt_ = 0.0;
      if (a.mu_b)
        for (int c = lane; c < a.Co; c += 64) bt_ += (double)kl_term(a.mu_b[c], softplus(a.rho_b[c]), a.pmu_b[c], a.psig_b[c]);
      bt_ = wave_sum(bt_);
      if (lane == 0) {
        float kl = (float)(t / (double)a.w_elems);
        if (a.mu_b) kl += (float)(bt_ / (double)a.Co);
        a.kl_out[0] = kl;
        __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

  // ---- the contraction: a wave = 64 channels x its 32 columns
  f32x16 acc[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int Q = 0; Q < 8; ++Q) {
    if (Q < nsteps) {
      const char* const wp = wbuf + Q * W_STEP + lh * W_HALF + (li ^ ((2 * Q + lh) & 7)) * 16;
      bf16x8 wf[TN][NP];
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) wf[i][p] = *reinterpret_cast<const bf16x8*>(wp + p * W_PIECE + i * 32 * 16);
      uint32_t ph[8], pm[8], pl[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) split_pieces(xr[Q][c], ph[c], pm[c], pl[c]);
      const uint4 h4 = make_uint4(pack_hi16(ph[1], ph[0]), pack_hi16(ph[3], ph[2]), pack_hi16(ph[5], ph[4]), pack_hi16(ph[7], ph[6]));
      const uint4 m4 = make_uint4(pack_hi16(pm[1], pm[0]), pack_hi16(pm[3], pm[2]), pack_hi16(pm[5], pm[4]), pack_hi16(pm[7], pm[6]));
      const uint4 l4 = make_uint4(pack_hi16(pl[1], pl[0]), pack_hi16(pl[3], pl[2]), pack_hi16(pl[5], pl[4]), pack_hi16(pl[7], pl[6]));
      const bf16x8 x0 = __builtin_bit_cast(bf16x8, h4), x1 = __builtin_bit_cast(bf16x8, m4), x2 = __builtin_bit_cast(bf16x8, l4);
#pragma unroll
      for (int i = 0; i < TN; ++i) {   // D[channel][column]: weights as A, columns as B; the six terms in every kernel's order
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], x0, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][1], x0, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], x1, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][2], x0, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][1], x1, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i][0], x2, acc[i], 0, 0, 0);
      }
    }
  }

  // ---- the slice's partial tile -> its slab: [column][64 channels], 16-byte stores (register r of acc[i] = channel i*32 + (r&3) +
  // 8*(r>>2) + 4*lh of column wave*32 + li: four consecutive channels per (r >> 2))
  // WRITE-THROUGH (sc1) 16-byte stores, drained by every wave, then ONE relaxed agent-scope ticket add -- no release fence: on this part
  // an agent-scope release writes back the XCD's whole L2, i.e. everybody's slabs (first version of this kernel: 100 us for the layer
  // the general kernel does in 57). The reducer reads the slabs with sc1 loads, which bypass its L1 (cdna_hip_programming.md,
  // Guideline 16: every handed-off byte stored sc1 and drained, every load of it sc1 -- no acquire needed).
  constexpr int SLAB = BN * kSkinnyCols * 4;   // bytes
  const __amdgpu_buffer_rsrc_t r_sl = __builtin_amdgcn_make_buffer_rsrc(a.sk_scratch + (long long)tile * NSL * (BN * kSkinnyCols), 0, NSL * SLAB, 0x00020000);
  {
    const int off = q * SLAB + ((wave * 32 + li) * BN + 4 * lh) * 4;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 v = make_float4(acc[i][4 * r4], acc[i][4 * r4 + 1], acc[i][4 * r4 + 2], acc[i][4 * r4 + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r_sl, off + (i * 32 + 8 * r4) * 4, 0, 16);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(&a.sk_tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == (unsigned)NSL - 1u ? 1 : 0;
    if (last) __hip_atomic_store(&a.sk_tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the workspace zeroed
    flag[0] = last;
  }
  // bias draw + output-stage constants of the tile's channels (only the reducer reads them; their loads were issued at the top)
  if (tid >= 64 && tid < 128) {
    float bv = 0.f;
    if (c_bias) {
      const int co = g * a.Cog + n0 + tid - 64;
      RngKey kb = key_w;
      kb.layer_tensor = layer_tensor_word(a.layer_id, 1);
      float z[4];
      philox_normal4(kb, sample, (uint32_t)(co >> 2), z);
      const int sel = co & 3;
      const float e = sel == 0 ? z[0] : sel == 1 ? z[1] : sel == 2 ? z[2] : z[3];
      bv = __fadd_rn(c_mub, __fmul_rn(softplus(c_rhob), e));
    }
    cst[tid - 64] = make_float4(bv, c_sc, c_sh, 0.f);
  }
  __syncthreads();
  if (!flag[0]) return;

  // ---- the reducer: every slab of the tile, in slice order; then the output stage. A thread owns 4 consecutive channels of a column.
  float* const out_s = a.out + (long long)s * a.out_elems;
  const float* const res_s = a.ep_res ? a.ep_res + (long long)s * a.ep_res_stride : nullptr;
  const bool relu = a.ep_relu != 0;
  const bool vec_out = (a.Co & 3) == 0 && (a.Cog & 3) == 0 && ((((uintptr_t)out_s | (uintptr_t)res_s) & 15u) == 0) && (a.ep_res_stride & 3) == 0 && (a.out_elems & 3) == 0;
  // A thread's 8 quads of a slab are fetched together, two slabs per trip (16 independent 16-byte loads in flight); the additions stay
  // in slice order per element.
  constexpr int EPT = BN * kSkinnyCols / 4 / kSkinnyThreads;   // 8 quads per thread
  auto ld_sl = [&](int byte_off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_sl, byte_off, 0, 16)); };   // aux 16: sc1
  float4 vsum[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) vsum[j] = ld_sl((tid + j * kSkinnyThreads) * 16);
  for (int k = 1; k < NSL; k += 2) {
    float4 w0[EPT], w1[EPT];
    const bool two = k + 1 < NSL;
    const int o0 = k * SLAB, o1 = (two ? k + 1 : k) * SLAB;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      w0[j] = ld_sl(o0 + (tid + j * kSkinnyThreads) * 16);
      w1[j] = ld_sl(o1 + (tid + j * kSkinnyThreads) * 16);
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      vsum[j].x = __fadd_rn(vsum[j].x, w0[j].x), vsum[j].y = __fadd_rn(vsum[j].y, w0[j].y), vsum[j].z = __fadd_rn(vsum[j].z, w0[j].z), vsum[j].w = __fadd_rn(vsum[j].w, w0[j].w);
      if (two) vsum[j].x = __fadd_rn(vsum[j].x, w1[j].x), vsum[j].y = __fadd_rn(vsum[j].y, w1[j].y), vsum[j].z = __fadd_rn(vsum[j].z, w1[j].z), vsum[j].w = __fadd_rn(vsum[j].w, w1[j].w);
    }
  }
  // output stage: the residual quads of the fast path in one batch ahead of the first store (out and residual may alias for all the compiler knows)
  float4 rq[EPT];
  if (res_s && vec_out) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = tid + j * kSkinnyThreads, col = e >> 4, c4 = (e & 15) * 4, bb = mt * kSkinnyCols + col;
      rq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bb < a.B && a.Cog - n0 - c4 >= 4) rq[j] = *reinterpret_cast<const float4*>(res_s + ((long long)bb * a.Co + g * a.Cog + n0 + c4));
    }
  }
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int e = tid + j * kSkinnyThreads;
    const int col = e >> 4, c4 = (e & 15) * 4;
    float4 v = vsum[j];
    const int bb = mt * kSkinnyCols + col;
    if (bb >= a.B) continue;
    const float4 k0 = cst[c4], k1 = cst[c4 + 1], k2 = cst[c4 + 2], k3 = cst[c4 + 3];
    v.x = __fadd_rn(__fmul_rn(__fadd_rn(v.x, k0.x), k0.y), k0.z), v.y = __fadd_rn(__fmul_rn(__fadd_rn(v.y, k1.x), k1.y), k1.z);
    v.z = __fadd_rn(__fmul_rn(__fadd_rn(v.z, k2.x), k2.y), k2.z), v.w = __fadd_rn(__fmul_rn(__fadd_rn(v.w, k3.x), k3.y), k3.z);
    const long long o = (long long)bb * a.Co + g * a.Cog + n0 + c4;
    const int nc = a.Cog - n0 - c4;   // channels of this quad that exist
    if (nc <= 0) continue;
    if (vec_out && nc >= 4) {
      if (res_s) v.x = __fadd_rn(v.x, rq[j].x), v.y = __fadd_rn(v.y, rq[j].y), v.z = __fadd_rn(v.z, rq[j].z), v.w = __fadd_rn(v.w, rq[j].w);
      if (relu) v.x = v.x < 0.f ? 0.f : v.x, v.y = v.y < 0.f ? 0.f : v.y, v.z = v.z < 0.f ? 0.f : v.z, v.w = v.w < 0.f ? 0.f : v.w;
      *reinterpret_cast<float4*>(out_s + o) = v;
    } else {
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int c = 0; c < 4 && c < nc; ++c) {
        float t = vv[c];
        if (res_s) t = __fadd_rn(t, res_s[o + c]);
        t = (relu && t < 0.f) ? 0.f : t;
        out_s[o + c] = t;
      }
    }
  }
}

}  // namespace bt
