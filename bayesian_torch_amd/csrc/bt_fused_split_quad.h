// Split-precision fused forward for layers with at most 4 input channels per group -- the 7x7 / stride-2 stems of the
// ResNets (3 -> 64 channels, 49 taps), optionally with the fused MaxPool2d(3, 2, 1) of the output stage.
//
// Same arithmetic as bt_fused_split.h (fp32 operands as exact bf16x3 pieces, 6 product terms on the bf16 matrix pipe, fp32
// accumulation) and the same draw stream (one Philox block = the 4 padded channels of one (row, tap)); what differs is the
// K packing, because an 8-channel octet would be 5/8 padding here:
//   * a k-group of 8 = TWO taps x 4 channels: one MFMA step covers FOUR taps (lanes 0-31: taps 4j, 4j+1; lanes 32-63: 4j+2,
//     4j+3) -- 49 taps = 13 steps instead of 25. Canonical K order of these layers: the active taps in groups of four.
//   * x patch: X[pixel][piece][4 ch] bf16 = 24 B per pixel, staged ONCE per tile (there is a single channel group), in the
//     room of both x buffers (96 KB = 4096 pixels: two 37x37 patches of a 7x7 / stride-2 stem on 16x16 outputs); an operand
//     fragment is two ds_read_b64 (one per tap of the lane half) per piece;
//   * W: as in bt_fused_split.h; a weight unit (row, tap) fills half a 16-byte slot. A stage is 5 steps = 20 taps; 49 taps
//     = 3 stages over the same staged patch (W double-buffered).
#pragma once
#include "bt_fused_split.h"

namespace bt {

constexpr int kQuadXBytes = 2 * split_x_bytes<512, 3>();  // 98,304 B: 4096 pixels of 24 B
constexpr int kQuadXBytesFlip = 38400;                     // Flipout: what two weight images leave: 1600 pixels of 24 B
template <bool FLIP>
constexpr int quad_lds_bytes() { return 2 * split_w_bytes<64, 3, FLIP>() + (FLIP ? kQuadXBytesFlip : kQuadXBytes) + kSplitMiscBytes; }

// FLIP (Flipout, <= 3 input channels): as bt_fused_split.h's -- two weight images (mu | sigma*eps), two accumulator sets, 4 consumer
// waves of 32 channels x 128 pixels (BM = 256), s_out in the output stage. The sign bits of a pixel's (<= 3) channels ride in
// the PADDING channel of its first piece (bits 0-2 of that bf16: a denormal that only ever meets the zero weights of the padding
// channel), so the patch stays 24 B per pixel and the 37x37 patch of a CIFAR stem fits beside the two weight images; a
// consumer expands them to sign masks and flips its x fragments in registers between the two contractions.
template <int NP, bool POOL, bool FLIP = false>
__global__ __launch_bounds__(512) void fused_split_quad_kernel(const FwdArgs a) {
  constexpr int BN = 64, BM = FLIP ? 256 : 512, kProducers = 256, kThreadsAll = 512, STEPS = kSplitSteps, TPS = 4 * STEPS;  // taps per stage
  constexpr int CWM = FLIP ? 2 : 4, CWN = 4 / CWM, WTM = BM / CWM, TN = BN / CWN / 32, TM = WTM / 32, NOP = FLIP ? 2 : 1;
  constexpr int PBQ = 8 * NP;  // bytes per pixel of the quad patch
  constexpr int W_BYTES = split_w_bytes<BN, NP, FLIP>(), W_OP = W_BYTES / NOP, XQ_BYTES = FLIP ? kQuadXBytesFlip : kQuadXBytes;
  constexpr int W_STEP = 2 * NP * BN * 16, W_HALF = NP * BN * 16, W_PIECE = BN * 16;
  // (the patch region holds XQ_BYTES / PBQ pixels: the host checks PCH against it)
  constexpr int SROWS = BN, SROW = BM + 4;
  static_assert(!FLIP || NP == 3, "Flipout: the exact split");
  static_assert((4 * BN + NOP * SROWS * SROW) * 4 <= 2 * W_BYTES + XQ_BYTES, "output staging fits the operand buffers");

  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  char* const wbuf = smem_c;                  // [2][W_BYTES]
  char* const xq = smem_c + 2 * W_BYTES;      // one patch, XQ_BYTES
  float* const smem = reinterpret_cast<float*>(smem_c);
  int4* const taptab = reinterpret_cast<int4*>(smem_c + 2 * W_BYTES + XQ_BYTES);
  double* const red = reinterpret_cast<double*>(smem_c + 2 * W_BYTES + XQ_BYTES + kMaxTaps * 16);
  int* const misc = reinterpret_cast<int*>(red + 12);
  int* const eofftab = reinterpret_cast<int*>(taptab + 64);  // [chunk][step][4 taps]: byte offset of the tap inside the patch (T <= 64: upper half of the tap table's room)
  (void)red;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long* const dbg_ = kStamps ? a.dbg : nullptr;  // diagnostic build only (make stamps): tools/stamps.py
  if (dbg_ && dbg_[201] && tid == 0) {  // per-workgroup timeline
    dbg_[256 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    dbg_[256 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
  }
  const bool stamp0 = dbg_ && blockIdx.x == 0 && tid == 0;
  if (stamp0) dbg_[210] = __builtin_amdgcn_s_memtime();
  const bool producer = wave >= 4;
  const int ptid = producer ? tid - 256 : tid;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & (CWM - 1), wn = (wave & 3) / CWM;

  int L = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, a.total_blocks));   // (reciprocals from the host: bt_fused_split_host.h)
  int Lq = udiv_inv(L, a.m_tiles, a.inv_m_tiles);
  const int mt = __builtin_amdgcn_readfirstlane(L - Lq * a.m_tiles);
  L = Lq, Lq = udiv_inv(L, a.S, a.inv_S);
  const int s = __builtin_amdgcn_readfirstlane(L - Lq * a.S);
  L = Lq, Lq = udiv_inv(L, a.n_tiles, a.inv_n_tiles);
  const int nt = __builtin_amdgcn_readfirstlane(L - Lq * a.n_tiles);
  const int g = __builtin_amdgcn_readfirstlane(Lq);
  const int n0 = nt * BN;
  const int t_NI = a.t_NI, t_R = a.t_R, t_Wt = a.t_Wt, RW = t_R * t_Wt, Mt = t_NI * RW;
  const int trest = udiv_inv(mt, a.n_bt, a.inv_n_bt);
  const int bt = __builtin_amdgcn_readfirstlane(mt - trest * a.n_bt);
  const int rt = __builtin_amdgcn_readfirstlane(udiv_inv(trest, a.n_ct, a.inv_n_ct)), ct = __builtin_amdgcn_readfirstlane(trest - rt * a.n_ct);
  const int b0 = bt * t_NI, r0 = rt * t_R, w0 = ct * t_Wt;
  const uint32_t inv_rw = RW > 1 ? (a.inv_rw ? a.inv_rw : inv32(RW)) : 0u;
  const uint32_t inv_wt = t_Wt > 1 ? (a.inv_wt ? a.inv_wt : inv32(t_Wt)) : 0u;
  auto col_decode = [&](int ml, int& b, int& ho, int& wo) -> bool {
    const int img = RW == 1 ? ml : (int)__umulhi((uint32_t)ml, inv_rw);
    const int rem = ml - img * RW;
    const int r = t_Wt == 1 ? rem : (int)__umulhi((uint32_t)rem, inv_wt);
    b = b0 + img, ho = r0 + r, wo = w0 + (rem - r * t_Wt);
    return ml < Mt && b < a.B && ho < a.Ho && wo < a.Wo;
  };
  const uint32_t sample = a.sample0 + (uint32_t)s;
  const int T = a.T, Cig = a.Cig;  // Cig <= 4: the packed tensors hold one quad per (row, tap)

  RngKey key_w;
  key_w.seed_lo = a.seed_lo;
  key_w.seed_hi = a.seed_hi;
  key_w.call = a.call + (a.call_base ? __builtin_nontemporal_load(a.call_base) : 0u);
  key_w.layer_tensor = layer_tensor_word(a.layer_id, 0);
  uint32_t skey_in = 0, skey_out = 0;  // Flipout sign streams (bt_fused_fwd.h)
  if constexpr (FLIP) {
    RngKey ks = key_w;
    ks.layer_tensor = layer_tensor_word(a.layer_id, 2);
    skey_in = sign_stream_key(ks, sample);
    ks.layer_tensor = layer_tensor_word(a.layer_id, 3);
    skey_out = sign_stream_key(ks, sample);
  }

  if (wave == 0) {  // active taps + their window
    bool act = false;
    int4 e = make_int4(0, 0, 0, 0);
    if (lane < T) {
      const int kh = udiv_inv(lane, a.KW, a.inv_kw), kw = lane - kh * a.KW;
      e = make_int4(0, kh * a.DH, kw * a.DW, lane);
      const int lo_h = a.PH - e.y, lo_w = a.PW - e.z;
      const int hc = lo_h > 0 ? div_small(lo_h + a.SH - 1, a.SH) : 0, wc = lo_w > 0 ? div_small(lo_w + a.SW - 1, a.SW) : 0;
      act = hc < a.Ho && hc * a.SH - lo_h < a.H && wc < a.Wo && wc * a.SW - lo_w < a.W;
    }
    const unsigned long long mask = __ballot(act);
    if (act) taptab[__popcll(mask & ((1ull << lane) - 1ull))] = e;
    int dy0 = act ? e.y : (1 << 20), dy1 = act ? e.y : -1, dx0 = act ? e.z : (1 << 20), dx1 = act ? e.z : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      dy0 = min(dy0, __shfl_xor(dy0, o, 64)), dy1 = max(dy1, __shfl_xor(dy1, o, 64));
      dx0 = min(dx0, __shfl_xor(dx0, o, 64)), dx1 = max(dx1, __shfl_xor(dx1, o, 64));
    }
    if (lane == 0) misc[0] = __popcll(mask), misc[2] = dy0, misc[3] = dy1, misc[4] = dx0, misc[5] = dx1;
  }
  __syncthreads();
  const int nA = __builtin_amdgcn_readfirstlane(misc[0]);
  const int dymin = __builtin_amdgcn_readfirstlane(misc[2]), dymax = __builtin_amdgcn_readfirstlane(misc[3]);
  const int dxmin = __builtin_amdgcn_readfirstlane(misc[4]), dxmax = __builtin_amdgcn_readfirstlane(misc[5]);
  const int ps_h = (dymax == dymin) ? 1 : a.SH, gs_h = (dymax == dymin) ? a.SH : 1;
  const int ps_w = (dxmax == dxmin) ? 1 : a.SW, gs_w = (dxmax == dxmin) ? a.SW : 1;
  const int PHt = (t_R - 1) * ps_h + (dymax - dymin) + 1, PWt = (t_Wt - 1) * ps_w + (dxmax - dxmin) + 1;
  const int PIMG = PHt * PWt, PCH = t_NI * PIMG;  // host: PCH <= XCAP
  const int x_lo = w0 * a.SW - a.PW + dxmin, y_lo = r0 * a.SH - a.PH + dymin;
  const int NS = (nA + TPS - 1) / TPS;  // stages = tap chunks (0: degenerate geometry, outputs are the bias alone)
  for (int i = tid; i < NS * TPS; i += kThreadsAll) {  // tap offsets; dead taps read tap 0 (valid data) against zero weights
    const int4 e = taptab[i < nA ? i : 0];
    eofftab[i] = ((e.y - dymin) * PWt + (e.z - dxmin)) * PBQ;
  }
  // (No LDS clear: every W slot of the steps a stage runs is written by its unit -- masked units write zeros -- and every
  //  pixel of the patch is, its halo as the zeros of out-of-range loads; dead columns read pixel 0.)

  const float* const xs = a.x + (long long)s * a.x_sample_stride;
  constexpr uint32_t kOOB = 0x80000000u;
  const int pk_bytes = a.Co * T * 4 * 4;  // Cig4 == 4
  const __amdgpu_buffer_rsrc_t r_mu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mu_pk), 0, pk_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sig_pk), 0, pk_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, (int)(a.x_elems * 4), 0x00020000);
  auto ldf = [](const __amdgpu_buffer_rsrc_t& r, uint32_t byte_off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0)); };
  auto ldf4 = [](const __amdgpu_buffer_rsrc_t& r, uint32_t byte_off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0)); };

  float* const bias0 = smem;
  float* const bias1 = smem + BN;  // Flipout: the sigma*eps part of the bias
  float* const osc = smem + 2 * BN;
  float* const osh = smem + 3 * BN;
  const bool kl_block = a.do_kl && (int)blockIdx.x < a.kl_slices;
  float* const out_s = a.out + (long long)s * a.out_elems;
  const float* const res_s = a.ep_res ? a.ep_res + (long long)s * a.ep_res_stride : nullptr;
  const bool relu = a.ep_relu != 0;
  __syncthreads();  // tap offsets

  // ---- read-out of the staged output tile (all BN channels), by every wave ------------------------------------------------
  auto readout_quads = [&](int t0) {
    constexpr int QROW = BM / 4, NQD = SROWS * QROW, U = 8;
    const float* const stage = smem + 4 * BN;
    for (int c0q = t0; c0q < NQD; c0q += kThreadsAll * U) {
      uint32_t oidx[U];
      bool okq[U];
      float4 v[U], r4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cr = c0q + kThreadsAll * u, c = cr < NQD ? cr : 0;
        const int row = c / QROW, m4 = c - row * QROW;
        int bq, hq, wq;
        const bool mok = col_decode(4 * m4, bq, hq, wq);
        okq[u] = cr < NQD && mok && n0 + row < a.Cog;
        oidx[u] = okq[u] ? (uint32_t)(((bq * a.Co + g * a.Cog + n0 + row) * a.Ho + hq) * a.Wo + wq) : 0u;
        v[u] = *reinterpret_cast<const float4*>(stage + row * SROW + 4 * m4);
      }
      if (res_s) {
#pragma unroll
        for (int u = 0; u < U; ++u) r4[u] = *reinterpret_cast<const float4*>(res_s + oidx[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
          v[u].x = __fadd_rn(v[u].x, r4[u].x), v[u].y = __fadd_rn(v[u].y, r4[u].y), v[u].z = __fadd_rn(v[u].z, r4[u].z), v[u].w = __fadd_rn(v[u].w, r4[u].w);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (relu) v[u].x = v[u].x < 0.f ? 0.f : v[u].x, v[u].y = v[u].y < 0.f ? 0.f : v[u].y, v[u].z = v[u].z < 0.f ? 0.f : v[u].z, v[u].w = v[u].w < 0.f ? 0.f : v[u].w;
        if (okq[u]) *reinterpret_cast<float4*>(out_s + oidx[u]) = v[u];
      }
    }
  };
  // Pooled read-out, fused MaxPool2d(3, 2, 1) of whole staged images with a power-of-two pooled width (bt_fused_fast.h): a wave
  // = Wp pooled columns x 64/Wp staged planes, walking the pooled rows; NaN wins, like torch's kernel.
  auto readout_pool = [&](int wv) {
    const float* const stage = smem + 4 * BN;
    auto nmax = [](float m, float v) { return (v > m || v != v) ? v : m; };
    const int Hp = a.ep_Hp, Wp = a.ep_Wp, PP = Hp * Wp;
    const int lwp = 31 - __clz(Wp), ppw = 64 >> lwp;
    const int px = lane & (Wp - 1), psub = lane >> lwp;
    const bool has_l = px > 0;
    const int xl = has_l ? 2 * px - 1 : 0;
    for (int q0 = wv * ppw; q0 < SROWS * t_NI; q0 += 8 * ppw) {
      const int q = q0 + psub;
      const int img = q / SROWS, row = q - img * SROWS;  // planes of one image are consecutive: channel fastest
      const int b = b0 + img;
      const bool ok = q < SROWS * t_NI && b < a.B && n0 + row < a.Cog;
      const float* const plane = stage + (ok ? row * SROW + img * RW : 0);
      float* const oplane = out_s + (ok ? (b * a.Co + g * a.Cog + n0 + row) * PP : 0);
#pragma unroll 2
      for (int py = 0; py < Hp; ++py) {
        float m = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int y = 2 * py - 1 + dy;
          const bool iny = (unsigned)y < (unsigned)a.Ho;
          const float* const rowp = plane + (iny ? y : 0) * a.Wo;
          const float2 cd = *reinterpret_cast<const float2*>(rowp + 2 * px);
          const float l = rowp[xl];
          m = nmax(m, (iny && has_l) ? l : -INFINITY);
          m = nmax(m, iny ? cd.x : -INFINITY);
          m = nmax(m, iny ? cd.y : -INFINITY);
        }
        if (relu) m = m < 0.f ? 0.f : m;  // max and ReLU commute
        if (ok) oplane[py * Wp + px] = m;
      }
    }
  };

  // Flipout: the two staged accumulator sets -> one (s_out from the hash stream of the output element, then the output-stage
  // constants, in the fp32 kernels' order), in place, by every wave; the read-outs above then see what the Reparameterization
  // consumers would have staged.
  auto combine_flip = [&](int t0) {
    constexpr int QROW = BM / 4, RSTEP = kThreadsAll / QROW, NIT = SROWS / RSTEP;
    static_assert(kThreadsAll % QROW == 0 && SROWS % RSTEP == 0, "a thread keeps its pixel quad");
    float* const stage = smem + 4 * BN;
    const int m4 = t0 % QROW, row0 = t0 / QROW;
    int bq, hq, wq;
    const bool mok = col_decode(4 * m4, bq, hq, wq);
    const int HoWo_ = a.Ho * a.Wo;
    const uint32_t obase = mok ? (uint32_t)(((bq * a.Co + g * a.Cog + n0 + row0) * a.Ho + hq) * a.Wo + wq) : 0u;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int row = row0 + k * RSTEP;
      float* const p0 = stage + row * SROW + 4 * m4;
      float4 v = *reinterpret_cast<const float4*>(p0);
      const float4 d = *reinterpret_cast<const float4*>(p0 + SROWS * SROW);
      const uint32_t oi = obase + (uint32_t)(k * RSTEP * HoWo_);
      const float sc = osc[row], sh = osh[row];
      v.x = __fadd_rn(__fmul_rn(__fadd_rn(v.x, __fmul_rn(d.x, hash_sign(skey_out, oi))), sc), sh);
      v.y = __fadd_rn(__fmul_rn(__fadd_rn(v.y, __fmul_rn(d.y, hash_sign(skey_out, oi + 1u))), sc), sh);
      v.z = __fadd_rn(__fmul_rn(__fadd_rn(v.z, __fmul_rn(d.z, hash_sign(skey_out, oi + 2u))), sc), sh);
      v.w = __fadd_rn(__fmul_rn(__fadd_rn(v.w, __fmul_rn(d.w, hash_sign(skey_out, oi + 3u))), sc), sh);
      *reinterpret_cast<float4*>(p0) = v;
    }
  };

// ---- the patch, once, by ALL 8 waves (the consumers have nothing to do before the first stage): a thread owns pixels
//      tid + 512 i; <= 4 channels per pixel, split on the way to LDS ----
  {
    const uint32_t inv_pimg = PIMG > 1 ? inv32(PIMG) : 0u;
    const uint32_t inv_pw = PWt > 1 ? inv32(PWt) : 0u;
    const int HWb = 4 * a.HW;
    constexpr int XB = 4;  // pixels in flight per thread
    for (int i0 = 0; i0 * kThreadsAll < PCH; i0 += XB) {
      float xv[XB][4];
      int pos_[XB];
      uint32_t xoff_[XB];
#pragma unroll
      for (int k = 0; k < XB; ++k) {
        const int pos = tid + kThreadsAll * (i0 + k);
        pos_[k] = pos;
        const int pp = pos < PCH ? pos : 0;
        const int img = PIMG == 1 ? pp : (int)__umulhi((uint32_t)pp, inv_pimg);
        const int rem = pp - img * PIMG;
        const int yy = PWt == 1 ? rem : (int)__umulhi((uint32_t)rem, inv_pw);
        const int xx = rem - yy * PWt;
        const int b = b0 + img, y = y_lo + yy * gs_h, x = x_lo + xx * gs_w;
        const bool ok = pos < PCH && b < a.B && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        const uint32_t off = ok ? (uint32_t)(4 * ((b * a.Ci + g * Cig) * a.HW + y * a.W + x)) : kOOB;
        xoff_[k] = ok ? off : 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[k][c] = ldf(r_x, (ok && c < Cig) ? off + (uint32_t)(c * HWb) : kOOB);
      }
#pragma unroll
      for (int k = 0; k < XB; ++k) {
        if (pos_[k] < PCH) {
          uint32_t ph[4], pm[4], pl[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) split_pieces(xv[k][c], ph[c], pm[c], pl[c]);
          if constexpr (FLIP) {  // s_in of channels 0..2 (hash of the element's offset in the sample's x) -> bits 16..18: the padding channel's lane
            uint32_t bits = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c)
              bits |= (__float_as_uint(hash_sign(skey_in, (xoff_[k] >> 2) + (uint32_t)(c * a.HW))) >> 31) << (16 + c);
            ph[3] = bits;
          }
          char* const dst = xq + pos_[k] * PBQ;
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(ph[1], ph[0]), pack_hi16(ph[3], ph[2]));
          *reinterpret_cast<uint2*>(dst + 8) = make_uint2(pack_hi16(pm[1], pm[0]), pack_hi16(pm[3], pm[2]));
          if constexpr (NP == 3) *reinterpret_cast<uint2*>(dst + 16) = make_uint2(pack_hi16(pl[1], pl[0]), pack_hi16(pl[3], pl[2]));
        }
      }
    }
  }
  if (stamp0) dbg_[212] = __builtin_amdgcn_s_memtime();   // patch staged (this thread's share)
  if (producer) {
    // (no producer priority here, unlike bt_fused_split.h's wide tiles: without it the Reparameterization stem measured 228 -> 222 us alone
    //  and 253.5 -> 247.7 us inside the cfg3 graph, the Flipout stem 474.0 -> 459.1 us inside cfg4's -- same box, BT_LIB_PATH A/B)
    // =================================================== PRODUCERS ===========================================================
    // ---- weights: unit u = (row n = u & 63, tap slot of the stage q = u >> 6): 4 sampled weights = one Philox block ----
    constexpr int UMAX = (BN * TPS + kProducers - 1) / kProducers;  // 5
    int l_off[UMAX];
    uint32_t u_co[UMAX];
#pragma unroll
    for (int i = 0; i < UMAX; ++i) {
      const int u = ptid + kProducers * i, n = u & (BN - 1), q = u >> 6;  // q < TPS
      const int st_ = q >> 2, hf = (q >> 1) & 1, sub = q & 1;
      const int co_g = n0 + n;
      u_co[i] = co_g < a.Cog ? (uint32_t)(g * a.Cog + co_g) : 0xFFFFFFFFu;
      l_off[i] = st_ * W_STEP + hf * W_HALF + (n ^ ((2 * st_ + hf) & 7)) * 16 + sub * 8;
    }
    float4 mu[UMAX], rs[UMAX];
    uint32_t ue[UMAX];  // draw index of the unit in this stage, or OOB
    auto load_w = [&](int st) {
#pragma unroll
      for (int i = 0; i < UMAX; ++i) {
        const int ai = st * TPS + ((ptid + kProducers * i) >> 6);
        const bool in = ai < nA && u_co[i] != 0xFFFFFFFFu;
        const int tap = taptab[ai < nA ? ai : 0].w;
        ue[i] = in ? (u_co[i] * (uint32_t)T + (uint32_t)tap) * 4u : (kOOB >> 2);
        mu[i] = ldf4(r_mu, in ? 4u * ue[i] : kOOB), rs[i] = ldf4(r_rs, in ? 4u * ue[i] : kOOB);
      }
    };
    if (NS > 0) load_w(0);
    for (int st = 0; st <= NS; ++st) {  // NS + 1 barriers, like the consumer arm
      if (st < NS) {
        char* const Wt = wbuf + (st & 1) * W_BYTES;
        float ep[UMAX][4];
#pragma unroll
        for (int i = 0; i < UMAX; ++i) philox_normal4(key_w, sample, ue[i] >> 2, ep[i]);
#pragma unroll
        for (int i = 0; i < UMAX; ++i) {
          const float m4[4] = {mu[i].x, mu[i].y, mu[i].z, mu[i].w}, s4[4] = {rs[i].x, rs[i].y, rs[i].z, rs[i].w};
          uint32_t wh[4], wm_[4], wl[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)  // masked units (dead taps, rows past Cog) loaded zeros: w = 0 -- and the slot IS written
            split_pieces(FLIP ? m4[j] : __fadd_rn(m4[j], __fmul_rn(s4[j], ep[i][j])), wh[j], wm_[j], wl[j]);
          char* const dst = Wt + l_off[i];
          *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(wh[1], wh[0]), pack_hi16(wh[3], wh[2]));
          *reinterpret_cast<uint2*>(dst + W_PIECE) = make_uint2(pack_hi16(wm_[1], wm_[0]), pack_hi16(wm_[3], wm_[2]));
          if constexpr (NP == 3) *reinterpret_cast<uint2*>(dst + 2 * W_PIECE) = make_uint2(pack_hi16(wl[1], wl[0]), pack_hi16(wl[3], wl[2]));
          if constexpr (FLIP) {  // second image: the perturbation sigma * eps
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pieces(__fmul_rn(s4[j], ep[i][j]), wh[j], wm_[j], wl[j]);
            *reinterpret_cast<uint2*>(dst + W_OP) = make_uint2(pack_hi16(wh[1], wh[0]), pack_hi16(wh[3], wh[2]));
            *reinterpret_cast<uint2*>(dst + W_OP + W_PIECE) = make_uint2(pack_hi16(wm_[1], wm_[0]), pack_hi16(wm_[3], wm_[2]));
            *reinterpret_cast<uint2*>(dst + W_OP + 2 * W_PIECE) = make_uint2(pack_hi16(wl[1], wl[0]), pack_hi16(wl[3], wl[2]));
          }
        }
        if (st + 1 < NS) load_w(st + 1);
      }
      __syncthreads();
    }
    if (ptid < BN) {  // bias draw + output-stage constants
      float bv = 0.f;
      const int co_g = n0 + ptid;
      if (a.mu_b && co_g < a.Cog) {
        const int co = g * a.Cog + co_g;
        RngKey kb = key_w;
        kb.layer_tensor = layer_tensor_word(a.layer_id, 1);
        float z[4];
        philox_normal4(kb, sample, (uint32_t)(co >> 2), z);
        const int sel = co & 3;
        const float e = sel == 0 ? z[0] : sel == 1 ? z[1] : sel == 2 ? z[2] : z[3];
        const float dl = __fmul_rn(softplus(a.rho_b[co]), e);
        bv = FLIP ? a.mu_b[co] : __fadd_rn(a.mu_b[co], dl);
        if constexpr (FLIP) bias1[ptid] = dl;
      } else if constexpr (FLIP) {
        bias1[ptid] = 0.f;
      }
      bias0[ptid] = bv;
      const bool cv = a.ep_scale && co_g < a.Cog;
      const int cs = cv ? g * a.Cog + co_g : 0;
      const float sc = a.ep_scale ? a.ep_scale[cs] : 1.f, sh = a.ep_shift ? a.ep_shift[cs] : 0.f;
      osc[ptid] = cv ? sc : 1.f;
      osh[ptid] = cv ? sh : 0.f;
    }
    __syncthreads();
    __syncthreads();
    if constexpr (FLIP) {
      combine_flip(tid);
      __syncthreads();
    }
    if constexpr (POOL) readout_pool(wave);
    else readout_quads(tid);
  } else {
    // =================================================== CONSUMERS ===========================================================
    long long kl_i = 0, kl_hi = 0;
    double kl_acc = 0.0;
    bool kl_v4 = false;
    if (kl_block) {
      long long chunk = (a.w_elems + a.kl_slices - 1) / a.kl_slices;
      chunk = (chunk + 3) & ~3ll;
      const long long lo = (long long)blockIdx.x * chunk;
      kl_hi = (lo + chunk < a.w_elems) ? lo + chunk : a.w_elems;
      kl_v4 = ((((uintptr_t)a.mu_w | (uintptr_t)a.rho_w | (uintptr_t)a.pmu_w | (uintptr_t)a.psig_w) & 15u) == 0);
      kl_i = lo + 4ll * ptid;
    }
    auto kl_group = [&]() {
      if (!(kl_v4 && kl_i + 3 < kl_hi)) return false;
      const float4 m4 = *reinterpret_cast<const float4*>(a.mu_w + kl_i), r4 = *reinterpret_cast<const float4*>(a.rho_w + kl_i);
      const float4 p4 = *reinterpret_cast<const float4*>(a.pmu_w + kl_i), q4 = *reinterpret_cast<const float4*>(a.psig_w + kl_i);
      const float t0 = kl_term(m4.x, softplus(r4.x), p4.x, q4.x) + kl_term(m4.y, softplus(r4.y), p4.y, q4.y);
      const float t1 = kl_term(m4.z, softplus(r4.z), p4.z, q4.z) + kl_term(m4.w, softplus(r4.w), p4.w, q4.w);
      kl_acc += (double)t0 + (double)t1;
      kl_i += 1024;
      return true;
    };
    auto kl_finish = [&]() {
      while (kl_group()) {}
      for (; kl_i < kl_hi; kl_i += 1024)
        for (int j = 0; j < 4; ++j)
          if (kl_i + j < kl_hi) kl_acc += (double)kl_term(a.mu_w[kl_i + j], softplus(a.rho_w[kl_i + j]), a.pmu_w[kl_i + j], a.psig_w[kl_i + j]);
      const double wsum = wave_sum(kl_acc);
      if (lane == 0) __hip_atomic_store(&a.slots[(int)blockIdx.x * 4 + wave], wsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto kl_ticket = [&]() {
      const int nslots = 4 * a.kl_slices;
      int last = 0;
      if (lane == 0) last = (__hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.kl_slices - 1u) ? 1 : 0;
      if (__builtin_amdgcn_readfirstlane(last)) {
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
          __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    };
    int colq[TM];  // byte offset of this lane's output pixel inside the patch, per 32-wide column group
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int ml = wm * WTM + j * 32 + li;
      int b, ho, wo;
      const bool live = col_decode(ml, b, ho, wo);
      colq[j] = live ? ((b - b0) * PIMG + (ho - r0) * ps_h * PWt + (wo - w0) * ps_w) * PBQ : 0;
    }
    int wlq[STEPS];
#pragma unroll
    for (int q = 0; q < STEPS; ++q) wlq[q] = lh * W_HALF + (li ^ ((2 * q + lh) & 7)) * 16 + wn * TN * 32 * 16;

    f32x16 acc[NOP][TN][TM];
#pragma unroll
    for (int o = 0; o < NOP; ++o)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[o][i][j][r] = 0.f;

    __syncthreads();  // patch and stage 0 staged
    if (stamp0) dbg_[0] = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < NS; ++st) {
      if (stamp0) dbg_[2 + 2 * st] = __builtin_amdgcn_s_memtime();
      const char* const Wt = wbuf + (st & 1) * W_BYTES;
      const int left = nA - st * TPS, nstep = left >= TPS ? STEPS : (left + 3) >> 2;
      // (step, column group) units in a software pipeline, as in bt_fused_split.h: the stage's tap offsets are fetched first,
      // the fragments of unit u+1 are read before the MFMAs of unit u.
      int eA[STEPS], eB[STEPS];  // this lane half's two taps of every step
#pragma unroll
      for (int q = 0; q < STEPS; ++q) eA[q] = eofftab[st * TPS + 4 * q + 2 * lh], eB[q] = eofftab[st * TPS + 4 * q + 2 * lh + 1];
      bf16x8 wf[2][NOP][TN][NP];
      uint2 xlo[2][NP], xhi[2][NP];
      auto read_w = [&](int q) {
#pragma unroll
        for (int o = 0; o < NOP; ++o)
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
              wf[q & 1][o][i][p] = *reinterpret_cast<const bf16x8*>(Wt + o * W_OP + wlq[q] + q * W_STEP + p * W_PIECE + i * 32 * 16);
      };
      auto read_x = [&](int u) {
        const int q = u / TM, j = u % TM;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          xlo[u & 1][p] = *reinterpret_cast<const uint2*>(xq + colq[j] + eA[q] + 8 * p);
          xhi[u & 1][p] = *reinterpret_cast<const uint2*>(xq + colq[j] + eB[q] + 8 * p);
        }
      };
      read_w(0);
      read_x(0);
#pragma unroll
      for (int q = 0; q < STEPS; ++q) {
        if (q < nstep) {  // uniform
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            const int u = q * TM + j;
            if (j + 1 < TM) {
              read_x(u + 1);
            } else if (q + 1 < STEPS) {
              if (q + 1 < nstep) {
                read_x(u + 1);
                read_w(q + 1);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 xf[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) xf[p] = __builtin_bit_cast(bf16x8, make_uint4(xlo[u & 1][p].x, xlo[u & 1][p].y, xhi[u & 1][p].x, xhi[u & 1][p].y));
#pragma unroll
            for (int o = 0; o < NOP; ++o) {
              if (o == 1) {  // Flipout: x o s_in. The sign bits of a pixel's channels 0..2 sit in bits 16..18 of its first piece's second dword
                const uint32_t mA = xlo[u & 1][0].y, mB = xhi[u & 1][0].y;
                const uint4 mk = make_uint4(((mA & 0x10000u) >> 1) | ((mA & 0x20000u) << 14), (mA & 0x40000u) >> 3,
                                            ((mB & 0x10000u) >> 1) | ((mB & 0x20000u) << 14), (mB & 0x40000u) >> 3);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                  uint4 t = __builtin_bit_cast(uint4, xf[p]);
                  t.x ^= mk.x, t.y ^= mk.y, t.z ^= mk.z, t.w ^= mk.w;
                  xf[p] = __builtin_bit_cast(bf16x8, t);
                }
              }
#pragma unroll
              for (int i = 0; i < TN; ++i) {
                acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[0], wf[q & 1][o][i][0], acc[o][i][j], 0, 0, 0);
                acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[0], wf[q & 1][o][i][1], acc[o][i][j], 0, 0, 0);
                acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[1], wf[q & 1][o][i][0], acc[o][i][j], 0, 0, 0);
                if constexpr (NP == 3) {
                  acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[0], wf[q & 1][o][i][2], acc[o][i][j], 0, 0, 0);
                  acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[1], wf[q & 1][o][i][1], acc[o][i][j], 0, 0, 0);
                  acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[2], wf[q & 1][o][i][0], acc[o][i][j], 0, 0, 0);
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (stamp0) dbg_[2 + 2 * st + 1] = __builtin_amdgcn_s_memtime();
      if (kl_block) kl_group();
      __syncthreads();
    }
    if (kl_block) kl_finish();
    if (stamp0) dbg_[1] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // bias / output-stage constants staged; KL partials published
    if (kl_block && wave == 0) kl_ticket();

    // ---- output stage: the whole tile through LDS (lane = one channel, registers 4q..4q+3 = 4 consecutive positions) ----
    float* const stage = smem + 4 * BN;
    float bsv[TN], scv[TN], shv[TN], b1v[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int co_l = (wn * TN + i) * 32 + li;
      bsv[i] = bias0[co_l], scv[i] = osc[co_l], shv[i] = osh[co_l];
      b1v[i] = FLIP ? bias1[co_l] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      float* const srow = stage + ((wn * TN + i) * 32 + li) * SROW + wm * WTM + 4 * lh;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
          if constexpr (FLIP) {  // both sets as they are (+ their bias parts): combine_flip joins them
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(acc[0][i][j][4 * q + e], bsv[i]);
            *reinterpret_cast<float4*>(srow + j * 32 + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(acc[NOP - 1][i][j][4 * q + e], b1v[i]);
            *reinterpret_cast<float4*>(srow + SROWS * SROW + j * 32 + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(__fmul_rn(__fadd_rn(acc[0][i][j][4 * q + e], bsv[i]), scv[i]), shv[i]);
            *reinterpret_cast<float4*>(srow + j * 32 + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
    __syncthreads();
    if constexpr (FLIP) {
      combine_flip(tid);
      __syncthreads();
    }
    if (stamp0) dbg_[121] = __builtin_amdgcn_s_memtime();
    if constexpr (POOL) readout_pool(wave);
    else readout_quads(tid);
    if (stamp0) dbg_[126] = __builtin_amdgcn_s_memtime();
  }
  if (dbg_ && dbg_[201] && tid == 0) {
    dbg_[256 + 4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    dbg_[256 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime() - dbg_[256 + 4 * blockIdx.x + 2];
  }
}

}  // namespace bt
