// Fast flavour of the fused forward (see bt_fused_fwd.h for the general one and the shared argument block).
//
// Same algorithm, specialised to what the convolution / linear layers of real models look like so that the producers'
// steady-state loop is a handful of instructions per element on state that lives in VGPRs:
//   * the stage shape (CC channels x NA active taps) is fixed, so the unit decode, the patch geometry and -- when all
//     active taps fit one stage (kh*kw <= 9) -- each unit's tap and the K-row table are computed ONCE; larger kernels
//     (7x7 stems) walk the taps in chunks of 9 over the same staged patch;
//   * convolutions stage x as an LDS patch (each input pixel loaded once per stage, zero halo included); tiles are
//     t_NI images x t_R rows x t_Wt columns (whole images, row bands of large maps, or one pixel x BM images); the patch is
//     filled by one of three staging modes, one per instantiation (XMODE below); Linear stages its row-major tile with
//     float4 loads;
//   * spatial outputs leave through an LDS-staged tile (whole 128-byte lines per store), optionally max-pooled (POOL).
// Everything else (patches that do not fit LDS, injected draws, unaligned Linear, absent parameter packs) runs the general
// kernel, which keeps the same order of accumulation: the two flavours agree bit for bit.
#pragma once
#include "bt_fused_fwd.h"

namespace bt {

// buffer_load_dwordx4 ... lds: lane l copies 16 bytes from (resource base + voffset[l] + soffset) to lds_wave_base + 16 * l;
// an out-of-range lane writes zeros, a masked lane writes nothing (tools/ubench/lds_dma.hip). Tracked by vmcnt.
__device__ __forceinline__ void lds_dma16(const __amdgpu_buffer_rsrc_t& r, float* lds_wave_base, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)  // the builtin exists in the device pass only
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voffset, soffset, 0, 0);
#endif
}

// XMODE: how the conv x patch reaches LDS. 0: one dword per patch word through registers (any geometry; Linear ignores it).
// 1: 16-byte row chunks copied global -> LDS directly (host flag x_rows; Reparameterization only). 2: 16-byte channel
// vectors for 1x1 / 2x2 input planes when the tile's tap window allows, else as 0. One mode per instantiation keeps each
// kernel's producer loop free of the other modes' registers and scalars.
// NPW: producer waves (4: one per SIMD beside its consumer wave; 8: two per SIMD -- for the narrow tiles, whose weight
// synthesis per MFMA is the largest and whose registers allow 12 waves per workgroup).
// POOL: the output stage ends in MaxPool2d(3, 2, 1) (bt_epilogue.pool). Its own instantiation, of the row-chunk kernels only,
// so that no other kernel carries the pooled read-out.
template <int BN, int BM, int CWN, bool FLIP, bool LINEAR, bool TRANS, bool INJ, int XMODE = 0, int NPW = 4, bool POOL = false>
__global__ __launch_bounds__(256 + 64 * NPW) void fused_fast_kernel(const FwdArgs a) {
  static_assert(!POOL || (XMODE == 1 && TRANS), "the pooled output stage lives in the row-chunk instantiations");
  constexpr int kProducers = 64 * NPW;  // (shadows the general kernel's constant)
  static_assert(NPW == 4 || (NPW == 8 && !LINEAR), "producer waves");
  static_assert(XMODE == 0 || !LINEAR, "x staging modes are for conv patches");
  static_assert(XMODE != 1 || !FLIP, "Flipout multiplies x by its signs on the way to LDS");
  constexpr int CWM = 4 / CWN;
  constexpr int WTN = BN / CWN, WTM = BM / CWM;
  constexpr int TN = WTN / 32, TM = WTM / 32;
  constexpr int WS = BN + 1, XS = BM + 1;
  constexpr int NW = FLIP ? 2 : 1;
  constexpr int W_WORDS = kBK * WS, X_WORDS = x_words<BM, FLIP>(), BUF_WORDS = NW * (W_WORDS + X_WORDS);
  static_assert(TN >= 1 && TM >= 1 && WTM * CWM == BM && WTN * CWN == BN, "tile shape");
  static_assert(!LINEAR || BM <= 256, "a Linear x tile is [k][BM]: only the patch form fits tiles wider than 256");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  int4* const taptab = reinterpret_cast<int4*>(smem + 2 * BUF_WORDS);
  double* const red = reinterpret_cast<double*>(smem + 2 * BUF_WORDS + kMaxTaps * 4);
  int* const misc = reinterpret_cast<int*>(red + 12);
  int* const rowtab = misc + 8;

  // s_memtime stage stamps (tools/stamps.py) exist only in a diagnostic build (make STAMPS=1): in the product build the
  // pointer is a constant null and every stamp, its predicate and its scalar registers fold away.
  unsigned long long* const dbg_ = kStamps ? a.dbg : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int ptid = producer ? tid - 256 : tid;
  const int li = lane & 31, lh = lane >> 5;
  const int cw = wave & 3, wn = cw / CWM, wm = cw % CWM;

  int L = xcd_remap(blockIdx.x, a.total_blocks);
  // Integer division runs on the vector ALU even for uniform operands; readfirstlane brings the tile coordinates back to
  // scalar registers so everything derived from them (base pointers, buffer descriptors, loop bounds) stays scalar.
  const int mt = __builtin_amdgcn_readfirstlane(L % a.m_tiles);
  L /= a.m_tiles;
  const int s = __builtin_amdgcn_readfirstlane(L % a.S);
  L /= a.S;
  const int nt = __builtin_amdgcn_readfirstlane(L % a.n_tiles);
  const int g = __builtin_amdgcn_readfirstlane(L / a.n_tiles);
  const int n0 = nt * BN;
  // Tile geometry (host: bt_fused_dispatch.h): a tile is t_NI images x t_R output rows x t_Wt output columns starting at
  // (b0, r0, w0). Whole images (small feature maps), row bands of one image (ImageNet-size maps) and single pixels x BM
  // images (pixel-major) are all this form; tile column ml = (img * t_R + r) * t_Wt + w, columns >= Mt are dead.
  const bool pix = a.pixel_major != 0;
  const int t_NI = a.t_NI, t_R = a.t_R, t_Wt = a.t_Wt, RW = t_R * t_Wt, Mt = t_NI * RW;
  const int bt = __builtin_amdgcn_readfirstlane(mt % a.n_bt), trest = mt / a.n_bt;
  const int ct = __builtin_amdgcn_readfirstlane(trest % a.n_ct), rt = __builtin_amdgcn_readfirstlane(trest / a.n_ct);
  const int b0 = bt * t_NI, r0 = rt * t_R, w0 = ct * t_Wt;
  const int m0 = b0;  // Linear: first row of the tile
  const uint32_t inv_rw = RW > 1 ? (uint32_t)((0x100000000ull + (unsigned)RW - 1) / (unsigned)RW) : 0u;  // exact for ml < 2^16
  const uint32_t inv_wt = t_Wt > 1 ? (uint32_t)((0x100000000ull + (unsigned)t_Wt - 1) / (unsigned)t_Wt) : 0u;
  auto col_decode = [&](int ml, int& b, int& ho, int& wo) -> bool {  // tile column -> output coordinates; false: dead column
    const int img = RW == 1 ? ml : (int)__umulhi((uint32_t)ml, inv_rw);
    const int rem = ml - img * RW;
    const int r = t_Wt == 1 ? rem : (int)__umulhi((uint32_t)rem, inv_wt);
    b = b0 + img, ho = r0 + r, wo = w0 + (rem - r * t_Wt);
    return ml < Mt && b < a.B && ho < a.Ho && wo < a.Wo;
  };
  const uint32_t sample = a.sample0 + (uint32_t)s;
  const int K = a.K, T = a.T, Cig = a.Cig;

  RngKey key_w;
  key_w.seed_lo = a.seed_lo;
  key_w.seed_hi = a.seed_hi;
  key_w.call = a.call + (a.call_base ? __builtin_nontemporal_load(a.call_base) : 0u);
  key_w.layer_tensor = layer_tensor_word(a.layer_id, 0);
  uint32_t skey_in = 0, skey_out = 0;
  if (FLIP && !INJ) {
    RngKey ks = key_w;
    ks.layer_tensor = layer_tensor_word(a.layer_id, 2);
    skey_in = sign_stream_key(ks, sample);
    ks.layer_tensor = layer_tensor_word(a.layer_id, 3);
    skey_out = sign_stream_key(ks, sample);
  }

  // ---- active taps of this tile + their window (wave 0) ----------------------------------------------------------------
  if (wave == 0) {
    bool act = false;
    int4 e = make_int4(0, 0, 0, 0);
    if (lane < T) {
      const int kh = lane / a.KW, kw = lane - kh * a.KW;
      e = make_int4(0, kh * a.DH, kw * a.DW, lane);
      if (LINEAR) {
        act = true;
      } else if (pix) {  // the tile is one output pixel (r0, w0)
        act = (unsigned)(r0 * a.SH - a.PH + e.y) < (unsigned)a.H && (unsigned)(w0 * a.SW - a.PW + e.z) < (unsigned)a.W;
      } else {
        const int lo_h = a.PH - e.y, lo_w = a.PW - e.z;
        const int hc = lo_h > 0 ? (lo_h + a.SH - 1) / a.SH : 0, wc = lo_w > 0 ? (lo_w + a.SW - 1) / a.SW : 0;
        act = hc < a.Ho && hc * a.SH - lo_h < a.H && wc < a.Wo && wc * a.SW - lo_w < a.W;
      }
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
  const int nA = __builtin_amdgcn_readfirstlane(misc[0]);  // 0 only for degenerate geometry (no tap ever reaches data): outputs are the bias alone

  // ---- stage shape: CC channels x nA taps <= kBK rows --------------------------------------------------------------------
  const int NA = nA < 9 ? nA : 9;                       // taps per stage
  const int n_ach = nA ? (nA + NA - 1) / NA : 1;        // tap chunks (1 unless kh*kw > 9)
  int CC = (LINEAR || nA <= 1) ? 32 : nA == 2 ? 16 : nA <= 4 ? 8 : 4;
  while (CC > 4 && CC / 2 >= Cig) CC >>= 1;
  const int Cig4 = (Cig + 3) & ~3;
  // patch geometry (see bt_fused_fwd.h)
  const int dymin = __builtin_amdgcn_readfirstlane(misc[2]), dymax = __builtin_amdgcn_readfirstlane(misc[3]);
  const int dxmin = __builtin_amdgcn_readfirstlane(misc[4]), dxmax = __builtin_amdgcn_readfirstlane(misc[5]);
  const int ps_h = (dymax == dymin) ? 1 : a.SH, gs_h = (dymax == dymin) ? a.SH : 1;
  // Row-chunk mode (host flag): patch rows are whole 16-byte chunks of the input rows, copied global -> LDS without passing
  // through registers; the patch keeps every input column (ps_w = stride) and starts at a multiple of 4.
  constexpr bool kRows = XMODE == 1, xrows = kRows;
  const int x_lo = w0 * a.SW - a.PW + dxmin;  // input column of the first tap of the first output column
  int xa = x_lo, nchk = 0;
  if (xrows) nchk = row_chunks(x_lo, x_lo + (t_Wt - 1) * a.SW + (dxmax - dxmin), a.W, &xa);
  const int xshift = x_lo - xa;
  const int ps_w = (xrows || dxmax != dxmin) ? a.SW : 1, gs_w = (xrows || dxmax != dxmin) ? 1 : a.SW;
  const int PHt = (t_R - 1) * ps_h + (dymax - dymin) + 1, PWt = xrows ? 4 * nchk : (t_Wt - 1) * ps_w + (dxmax - dxmin) + 1;
  const int PIMG = PHt * PWt, PCH = t_NI * PIMG + (xrows ? 4 : 0);
  if (!LINEAR)
    while (CC > 4 && CC * PCH > X_WORDS) CC >>= 1;  // the host guaranteed 4 * PCH <= X_WORDS
  const int lcc = 31 - __clz(CC);
  // Channel-vector staging (XMODE 2) when this tile's tap window is the whole (tiny) input plane. The channel planes of
  // the LDS patch are then PST words apart, padded so the vectors' LDS writes spread over all banks.
  const int y_lo = r0 * a.SH - a.PH + dymin;  // input row of the patch origin
  bool cvecA = XMODE == 2 && a.HW == 1 && PIMG == 1 && (Cig & 3) == 0;
  bool cvecB = XMODE == 2 && a.HW == 4 && a.W == 2 && PIMG == 4 && PWt == 2 && x_lo == 0 && y_lo == 0 && gs_h == 1 && gs_w == 1;
  int PST = PCH;
  if (cvecA) PST = PCH + ((33 - (PCH & 31)) & 31);  // == 1 (mod 32): lanes = (image, channel quad) write 4 planes
  if (cvecB) PST = PCH + ((36 - (PCH & 31)) & 31);  // == 4 (mod 32), stays a multiple of 4: lanes = (image, channel) write 16 bytes
  if (CC * PST > X_WORDS) cvecA = cvecB = false, PST = PCH;
  const bool cvec = cvecA || cvecB;
  const int KC = NA << lcc;  // K rows of a full stage (a multiple of 4, <= 36)
  const int NS = nA ? n_ach * ((Cig + CC - 1) / CC) : 0;

  // K-row table of the x tile: consumers address  Xbuf[rowtab[k row] + colbase[lane]]. One table when every stage has
  // the same taps (n_ach == 1), else rewritten per stage by the producers.
  auto write_rowtab = [&](int slot, int a0, int rows, int t) {
    if (t < rows) {
      int off = t * XS;
      if (!LINEAR) {
        const int4 e = taptab[a0 + (t >> lcc)];
        off = (t & (CC - 1)) * PST + (e.y - dymin) * PWt + (e.z - dxmin) + xshift;
      }
      rowtab[slot * 40 + t] = off;
    }
  };
  if (n_ach == 1) write_rowtab(0, 0, KC, tid);

  const float* const xs = a.x + (long long)s * a.x_sample_stride;

  // Buffer resources: 32-bit byte offsets against a scalar base, and the hardware range check returns 0 for an offset
  // past the end -- a masked element is a load at kOOB, no select afterwards. (The host routes tensors of 2^29 elements
  // or more to the general kernel, so valid offsets stay below kOOB.)
  constexpr uint32_t kOOB = 0x80000000u;
  const int pk_bytes = a.Co * T * Cig4 * 4;  // packed parameter tensors [Co][T][Cig4]
  const __amdgpu_buffer_rsrc_t r_mu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mu_pk), 0, pk_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.sig_pk), 0, pk_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, (int)(a.x_elems * 4), 0x00020000);
  auto ldf = [](const __amdgpu_buffer_rsrc_t& r, uint32_t byte_off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0)); };
  auto ldf4 = [](const __amdgpu_buffer_rsrc_t& r, uint32_t byte_off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0)); };

  float* const wbuf = smem;                       // [2][NW][W_WORDS]  by stage parity
  float* const xbuf = smem + 2 * NW * W_WORDS;    // [2][NW][X_WORDS]  by channel-chunk parity (tap chunks share the patch)
  float* const bias0 = smem;
  float* const bias1 = smem + BN;
  float* const osc = smem + 2 * BN;
  float* const osh = smem + 3 * BN;
  const bool kl_block = a.do_kl && (int)blockIdx.x < a.kl_slices;

  // ---- output side, shared by both arms -------------------------------------------------------------------------------------
  float* const out_s = a.out + (long long)s * a.out_elems;
  const float* const res_s = a.ep_res ? a.ep_res + (long long)s * a.ep_res_stride : nullptr;
  const bool relu = a.ep_relu != 0;
  // Read-out of one staged pass of the output tile (see the consumers' output stage): quad c = (channel row, 4 positions),
  // consecutive threads on consecutive quads of a row -> whole 128-byte lines per store and per residual load. EVERY thread
  // of the workgroup takes part (the producers have nothing else left to do). Batches of U quads: all residual loads and
  // LDS reads of a batch are issued (at clamped addresses, no branches) before the first use.
  // (Only in the row-chunk instantiations -- the wide spatial tiles, where the output stage is 8 % of a workgroup's life;
  // elsewhere the extra code in the producer arm costs the main loop more than the read-out gains.)
  constexpr bool kReadoutAll = XMODE == 1;
  auto readout_quads = [&](int i, int t0) {
    constexpr int SROW = BM + 4, SROWS = 32 * CWN, QROW = BM / 4, NQ = SROWS * QROW, NT = kReadoutAll ? 256 + kProducers : 256;
    constexpr int NITc = (NQ + NT - 1) / NT, U = NITc < 8 ? NITc : 8;
    const float* const stage = smem + 4 * BN;
    for (int c0q = t0; c0q < NQ; c0q += NT * U) {
      uint32_t oidx[U];
      bool okq[U];
      float4 v[U], r4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cr = c0q + NT * u, c = cr < NQ ? cr : 0;
        const int row = c / QROW, m4 = c - row * QROW;
        int bq, hq, wq;
        const bool mok = col_decode(4 * m4, bq, hq, wq);
        const int co_l = (row >> 5) * WTN + i * 32 + (row & 31);
        okq[u] = cr < NQ && mok && n0 + co_l < a.Cog;
        oidx[u] = okq[u] ? (uint32_t)(((bq * a.Co + g * a.Cog + n0 + co_l) * a.Ho + hq) * a.Wo + wq) : 0u;
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

  // Pooled read-out of one staged pass, fused MaxPool2d(3, 2, 1), for power-of-two pooled widths (the 32x32-input stem:
  // 16 -> 8): a wave = Wp pooled columns x 64/Wp staged planes (consecutive channels: their LDS rows are 4 banks apart),
  // walking the pooled rows. Window geometry is lane state decoded once; a window row is one 8-byte read (columns 2px,
  // 2px+1) plus the left neighbour. NaN wins, like torch's kernel. Shared by all waves where the quads read-out is.
  const bool pool_pow2 = POOL && (a.ep_Wp & (a.ep_Wp - 1)) == 0 && a.ep_Wp >= 4 && a.ep_Wp <= 16;
  auto readout_pool = [&](int i, int wv) {
    constexpr int SROW = BM + 4, SROWS = 32 * CWN, NWV = kReadoutAll ? 4 + NPW : 4;
    const float* const stage = smem + 4 * BN;
    auto nmax = [](float m, float v) { return (v > m || v != v) ? v : m; };
    const int Hp = a.ep_Hp, Wp = a.ep_Wp, PP = Hp * Wp;
    const int lwp = 31 - __clz(Wp), ppw = 64 >> lwp;
    const int px = lane & (Wp - 1), psub = lane >> lwp;
    const bool has_l = px > 0;
    const int xl = has_l ? 2 * px - 1 : 0;
    for (int q0 = wv * ppw; q0 < SROWS * t_NI; q0 += NWV * ppw) {
      const int q = q0 + psub;
      const int img = q / SROWS, row = q - img * SROWS;  // planes of one image are consecutive: channel fastest
      const int co_l = (row >> 5) * WTN + i * 32 + (row & 31), b = b0 + img;
      const bool ok = q < SROWS * t_NI && b < a.B && n0 + co_l < a.Cog;
      const float* const plane = stage + (ok ? row * SROW + img * RW : 0);
      float* const oplane = out_s + (ok ? (b * a.Co + g * a.Cog + n0 + co_l) * PP : 0);
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

  if (producer) {
    // Producer instructions win issue arbitration over the consumer wave of the same SIMD: their VALU work slots in
    // between MFMAs instead of waiting for the consumer to stall.
    if (!dbg_ || dbg_[200] == 0) __builtin_amdgcn_s_setprio(3);
    // =================================================== PRODUCERS ===========================================================
    // ---- per-thread state, decoded once ----
    constexpr int UMAX = (BN * 9 + kProducers - 1) / kProducers;
    const int ncq = CC >> 2, lncq = lcc - 2;
    const int nunits = BN * ncq * NA;
    uint32_t e_off[UMAX];  // draw index of (unit, channel chunk 0[, tap when hoisted]) == element offset in the packed tensors
    int l_off[UMAX], c_lim[UMAX], u_ai[UMAX];  // LDS offset of element 0; channels left in this quad; tap slot
    {
      const uint32_t inv_na = NA > 1 ? (uint32_t)((0x100000000ull + (unsigned)NA - 1) / (unsigned)NA) : 0u;
#pragma unroll
      for (int i = 0; i < UMAX; ++i) {
        const int u = ptid + kProducers * i;
        const int uu = u < nunits ? u : 0;
        const int tq = NA <= 1 ? uu : (int)__umulhi((uint32_t)uu, inv_na);
        const int ai = uu - tq * (NA > 0 ? NA : 1);
        const int cq = tq & (ncq - 1), r = tq >> lncq;
        const int tap = (LINEAR || n_ach > 1) ? 0 : taptab[ai].w;  // several tap chunks: the tap is added per stage
        u_ai[i] = ai;
        const int co_g = n0 + r;
        const bool rv = u < nunits && co_g < a.Cog;
        const uint32_t co = (uint32_t)(g * a.Cog + (rv ? co_g : 0));
        e_off[i] = (co * (uint32_t)T + (uint32_t)tap) * (uint32_t)Cig4 + (uint32_t)(4 * cq);
        l_off[i] = ((ai << lcc) + 4 * cq) * WS + r;
        // 0: slot exists but holds zeros (row outside Cog, or a channel quad past Cig4 when the chunk is wider than the channel
        // count: those K rows must be written too -- a stale LDS word times a zero activation is not a zero if it is a NaN); -1: no slot
        c_lim[i] = rv ? (Cig4 - 4 * cq > 0 ? Cig4 - 4 * cq : 0) : (u < nunits ? 0 : -1);
      }
    }
    const int wave_u0 = __builtin_amdgcn_readfirstlane(ptid & ~63);
    // x patch: this thread owns plane positions ptid + 256*i (all channels of a stage)
    constexpr int PPOS = (X_WORDS / 4 + kProducers - 1) / kProducers;
    constexpr int QPOS = (X_WORDS / 16 + kProducers - 1) / kProducers;  // row-chunk mode: 16-byte chunks of one channel plane
    int p_off[XMODE == 1 ? 1 : PPOS];
    int q_off[kRows ? QPOS : 1];
    int cv_l[XMODE == 2 ? PPOS : 1];            // channel-vector mode: LDS word of the vector's first element, -1: no slot
    const int cv_c = cvecA ? (ptid & ((CC >> 2) - 1)) : (ptid & (CC - 1));  // this thread's channel quad / channel inside a stage
    const int nvec = cvecA ? t_NI * (CC >> 2) : t_NI * CC;                 // 16-byte vectors per stage
    const int NCHK = t_NI * PHt * nchk + 1;     // chunks per channel plane, the closing spare included
    if constexpr (xrows) {
      const int per_img = PHt * nchk;
      const uint32_t inv_img = (uint32_t)((0x100000000ull + (unsigned)per_img - 1) / (unsigned)per_img);
      const uint32_t inv_n = (uint32_t)((0x100000000ull + (unsigned)nchk - 1) / (unsigned)nchk);
#pragma unroll
      for (int i = 0; i < (kRows ? QPOS : 1); ++i) {
        const int q = ptid + kProducers * i;
        const int qq = q < NCHK - 1 ? q : 0;
        const int img = per_img == 1 ? qq : (int)__umulhi((uint32_t)qq, inv_img);
        const int rem = qq - img * per_img;
        const int yy = nchk == 1 ? rem : (int)__umulhi((uint32_t)rem, inv_n);
        const int b = b0 + img, y = y_lo + yy * gs_h, x = xa + 4 * (rem - yy * nchk);
        const bool ok = q < NCHK - 1 && b < a.B && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        q_off[i] = ok ? 4 * ((b * a.Ci + g * Cig) * a.HW + y * a.W + x) : (int)kOOB;  // halo chunks read (and write) zeros
      }
    } else if (XMODE == 2 && cvec) {
      // Tiny input planes: one 16-byte load covers 4 channels of a 1x1 input (cvecA) or the whole 2x2 plane of one channel
      // (cvecB). Consecutive lanes take consecutive channels of ONE image -- contiguous in memory -- instead of the same
      // channel of 64 images (one 4-byte piece out of 64 different lines per load instruction).
      if constexpr (XMODE == 2) {
#pragma unroll
        for (int i = 0; i < PPOS; ++i) {
          const int e = ptid + kProducers * i;
          const int img = cvecA ? e >> (lcc - 2) : e >> lcc;
          const int b = b0 + img;
          const bool ok = img < t_NI && b < a.B;
          p_off[i] = ok ? (cvecA ? 4 * (b * a.Ci + g * Cig + 4 * cv_c) : 16 * (b * a.Ci + g * Cig + cv_c)) : (int)kOOB;
          cv_l[i] = img < t_NI ? (cvecA ? 4 * cv_c * PST + img : cv_c * PST + 4 * img) : -1;
        }
      }
    } else if (!LINEAR) {
      const uint32_t inv_pimg = (uint32_t)((0x100000000ull + (unsigned)PIMG - 1) / (unsigned)PIMG);
      const uint32_t inv_pw = (uint32_t)((0x100000000ull + (unsigned)PWt - 1) / (unsigned)PWt);
#pragma unroll
      for (int i = 0; i < (XMODE == 1 ? 1 : PPOS); ++i) {
        const int pos = ptid + kProducers * i;
        const int pp = pos < PCH ? pos : 0;
        const int img = PIMG == 1 ? pp : (int)__umulhi((uint32_t)pp, inv_pimg);
        const int rem = pp - img * PIMG;
        const int yy = PWt == 1 ? rem : (int)__umulhi((uint32_t)rem, inv_pw);
        const int xx = rem - yy * PWt;
        const int b = b0 + img, y = y_lo + yy * gs_h, x = x_lo + xx * gs_w;
        const bool ok = pos < PCH && b < a.B && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        p_off[i] = ok ? 4 * ((b * a.Ci + g * Cig) * a.HW + y * a.W + x) : (int)kOOB;  // bytes; halo / outside the tile: reads 0
      }
    }

    auto produce = [&](auto LCCc, int st) {
      constexpr int LCC = decltype(LCCc)::value, CCs = 1 << LCC;
      const int cch = n_ach == 1 ? st : st / n_ach, ach = st - cch * n_ach;
      const int a0 = ach * NA, na_s = (nA - a0) < NA ? (nA - a0) : NA;
      float* const Wt0 = wbuf + (st & 1) * NW * W_WORDS;
      float* const Wt1 = Wt0 + W_WORDS;
      float* const Xt0 = xbuf + (cch & 1) * NW * X_WORDS;
      float* const Xt1 = Xt0 + X_WORDS;
      const int c0 = cch << LCC;
      uint32_t tap_e[UMAX];  // per-stage tap part of the draw index (zero when the tap is hoisted)
#pragma unroll
      for (int i = 0; i < UMAX; ++i) tap_e[i] = 0;
      if (n_ach > 1) {
#pragma unroll
        for (int i = 0; i < UMAX; ++i) {
          const int tap = taptab[a0 + (u_ai[i] < na_s ? u_ai[i] : 0)].w;
          tap_e[i] = (uint32_t)(tap * Cig4);
        }
        write_rowtab(st & 1, a0, na_s << LCC, ptid);
      }
      // ---- loads: weights ----
      static_assert(!INJ, "the fast flavour generates its draws on chip");
      float mu[UMAX][4], rs[UMAX][4], ep[UMAX][4];
#pragma unroll
      for (int i = 0; i < UMAX; ++i) {
        if (i == 0 || wave_u0 + kProducers * i < nunits) {  // wave-uniform
          // one 16-byte load per tensor: the packed layout puts the unit's 4 channels at the offset of its draw index
          const uint32_t sb = (u_ai[i] < na_s && c0 < c_lim[i]) ? 4u * (e_off[i] + tap_e[i] + (uint32_t)c0) : kOOB;
          const float4 m4 = ldf4(r_mu, sb), r4 = ldf4(r_rs, sb);
          mu[i][0] = m4.x, mu[i][1] = m4.y, mu[i][2] = m4.z, mu[i][3] = m4.w;
          rs[i][0] = r4.x, rs[i][1] = r4.y, rs[i][2] = r4.z, rs[i][3] = r4.w;
        }
      }
      const bool pst = dbg_ && blockIdx.x == 0 && tid == 256 && st == 3;
      if (pst) dbg_[251 - 11] = __builtin_amdgcn_s_memtime();
      // ---- loads: activations ----
      constexpr int RP = kProducers / 8;
      constexpr int PC = LINEAR ? 1 : (X_WORDS / CCs + kProducers - 1) / kProducers;
      constexpr int NXR = LINEAR ? ((BM + RP - 1) / RP) * 4 : (PC * CCs > 4 * PPOS ? PC * CCs : 4 * PPOS);
      float xv[NXR];
      uint32_t xo[FLIP ? NXR : 1];
      if constexpr (LINEAR) {
        const int kq = ptid & 7, mr = ptid >> 3;
#pragma unroll
        for (int p = 0; p < (BM + RP - 1) / RP; ++p) {
          const int rl = mr + p * RP;
          const int m = m0 + rl, k = c0 + 4 * kq;
          const uint32_t off = (uint32_t)m * (uint32_t)K + (uint32_t)k;
          const bool in = rl < BM && m < a.M && k < K;
          const float4 x4 = ldf4(r_x, in ? 4u * off : kOOB);
          xv[4 * p] = x4.x, xv[4 * p + 1] = x4.y, xv[4 * p + 2] = x4.z, xv[4 * p + 3] = x4.w;
          if constexpr (FLIP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xo[4 * p + j] = off + j;
          }
        }
      } else if constexpr (xrows) {
        {
          if (ach == 0) {  // 16-byte row chunks, global -> LDS directly: lane l of a wave fills chunk (wave's first chunk + l)
#pragma unroll
            for (int c = 0; c < CCs; ++c) {
              const bool chan = c0 + c < Cig;  // uniform; a padded channel's plane is filled with zeros
              const int soff = chan ? 4 * (c0 + c) * a.HW : 0;
#pragma unroll
              for (int i = 0; i < QPOS; ++i) {
                if (wave_u0 + kProducers * i < NCHK) {  // wave-uniform
                  if (ptid + kProducers * i < NCHK)
                    lds_dma16(r_x, Xt0 + c * PST + 4 * (wave_u0 + kProducers * i), chan ? q_off[i] : (int)kOOB, soff);
                }
              }
            }
          }
        }
      } else if (XMODE == 2 && cvec) {
        if (ach == 0) {
          const int cbyte = c0 * (cvecA ? 4 : 16);
          const bool chan = c0 + (cvecA ? 4 * cv_c : cv_c) < Cig;  // padded channels of the last stage read zeros
#pragma unroll
          for (int i = 0; i < PPOS; ++i) {
            if (kProducers * i < nvec) {  // uniform
              const uint32_t off = (chan && p_off[i] != (int)kOOB) ? (uint32_t)(p_off[i] + cbyte) : kOOB;
              const float4 x4 = ldf4(r_x, off);
              xv[4 * i] = x4.x, xv[4 * i + 1] = x4.y, xv[4 * i + 2] = x4.z, xv[4 * i + 3] = x4.w;

            }
          }
        }
      } else if (ach == 0) {  // later tap chunks reuse the staged patch
        const int c0HWb = 4 * c0 * a.HW, HWb = 4 * a.HW;
#pragma unroll
        for (int i = 0; i < PC; ++i) {
          if (kProducers * i < PCH) {  // uniform
#pragma unroll
            for (int c = 0; c < CCs; ++c) {
              const uint32_t off = (c0 + c < Cig) ? (uint32_t)(p_off[i] + c0HWb + c * HWb) : kOOB;  // channel test is uniform
              xv[i * CCs + c] = ldf(r_x, off);

            }
          }
        }
      }
      if (pst) dbg_[251] = __builtin_amdgcn_s_memtime();
      // ---- draws (no load feeds them) ----
      if constexpr (!INJ) {
#pragma unroll
        for (int i = 0; i < UMAX; ++i)
          if (i == 0 || wave_u0 + kProducers * i < nunits)  // wave-uniform only: masked lanes cost nothing, and without per-lane
            philox_normal4(key_w, sample, (e_off[i] + tap_e[i] + (uint32_t)c0) >> 2, ep[i]);  // branches the units' chains interleave
      }
      if (pst) dbg_[252] = __builtin_amdgcn_s_memtime();
      // ---- sampled weights -> LDS ----
#pragma unroll
      for (int i = 0; i < UMAX; ++i) {
        if ((i == 0 || wave_u0 + kProducers * i < nunits) && c_lim[i] >= 0 && u_ai[i] < na_s) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // masked units loaded zeros (mu = sigma = 0) and padded channels are zero in the pack: no select needed
            const float dl = __fmul_rn(rs[i][j], (c0 < c_lim[i]) ? ep[i][j] : 0.f);
            Wt0[l_off[i] + j * WS] = FLIP ? mu[i][j] : __fadd_rn(mu[i][j], dl);
            if (FLIP) Wt1[l_off[i] + j * WS] = dl;
          }
        }
      }
      if (pst) dbg_[253] = __builtin_amdgcn_s_memtime();
      // ---- activations -> LDS ----
      if constexpr (LINEAR) {
        const int kq = ptid & 7, mr = ptid >> 3;
#pragma unroll
        for (int p = 0; p < (BM + RP - 1) / RP; ++p) {
          const int rl = mr + p * RP;
          if (rl < BM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int idx = (4 * kq + j) * XS + rl;
              const float v = xv[4 * p + j];
              Xt0[idx] = v;
              if (FLIP) Xt1[idx] = __fmul_rn(v, hash_sign(skey_in, xo[4 * p + j]));
            }
          }
        }
      } else if (XMODE == 2 && cvec) {
        if (ach == 0) {
#pragma unroll
          for (int i = 0; i < (XMODE == 2 ? PPOS : 1); ++i) {
            if (kProducers * i < nvec && cv_l[i] >= 0) {
              const uint32_t xo0 = (uint32_t)(p_off[i] + c0 * (cvecA ? 4 : 16)) >> 2;  // sign index of the vector's first element
              if (cvecA) {  // 4 channels of one image: 4 planes
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  Xt0[cv_l[i] + j * PST] = xv[4 * i + j];
                  if (FLIP) Xt1[cv_l[i] + j * PST] = __fmul_rn(xv[4 * i + j], hash_sign(skey_in, xo0 + j));
                }
              } else {  // the 2x2 plane of one channel: 4 consecutive patch words
                *reinterpret_cast<float4*>(Xt0 + cv_l[i]) = make_float4(xv[4 * i], xv[4 * i + 1], xv[4 * i + 2], xv[4 * i + 3]);
                if (FLIP)
                  *reinterpret_cast<float4*>(Xt1 + cv_l[i]) =
                      make_float4(__fmul_rn(xv[4 * i], hash_sign(skey_in, xo0)), __fmul_rn(xv[4 * i + 1], hash_sign(skey_in, xo0 + 1)),
                                  __fmul_rn(xv[4 * i + 2], hash_sign(skey_in, xo0 + 2)), __fmul_rn(xv[4 * i + 3], hash_sign(skey_in, xo0 + 3)));
              }
            }
          }
        }
      } else if (ach == 0 && XMODE != 1) {
#pragma unroll
        for (int i = 0; i < PC; ++i) {
          const int pos = ptid + kProducers * i;
          if (pos < PCH) {
#pragma unroll
            for (int c = 0; c < CCs; ++c) {
              const float v = xv[i * CCs + c];
              Xt0[c * PST + pos] = v;
              // (the sign index = the element's index in the sample's x, rebuilt from the offsets instead of carried in registers;
              //  halo cells hold 0 and take whatever sign a wild index hashes to)
              if (FLIP) Xt1[c * PST + pos] = __fmul_rn(v, hash_sign(skey_in, (uint32_t)(p_off[i] + 4 * (c0 + c) * a.HW) >> 2));
            }
          }
        }
      }
    };

    const bool stamp = dbg_ && blockIdx.x == 0 && tid == 256;
    auto run_stages = [&](auto LCCc) {  // one loop per channel-chunk width: each carries only its own loop invariants
      for (int st = 0; st <= NS; ++st) {  // NS + 1 barriers, like the consumer arm
        if (stamp && st < 60) dbg_[128 + 2 * st] = __builtin_amdgcn_s_memtime();
        if (st < NS) produce(LCCc, st);
        if (stamp && st < 60) dbg_[128 + 2 * st + 1] = __builtin_amdgcn_s_memtime();
        __syncthreads();
      }
    };
    switch (lcc) {
      case 2: run_stages(std::integral_constant<int, 2>{}); break;
      case 3: run_stages(std::integral_constant<int, 3>{}); break;
      case 4: run_stages(std::integral_constant<int, 4>{}); break;
      default: run_stages(std::integral_constant<int, 5>{}); break;
    }
    // bias draw + output-stage constants of this workgroup's channels
    if (ptid < BN) {
      float b0 = 0.f, b1 = 0.f;
      const int co_g = n0 + ptid;
      if (a.mu_b && co_g < a.Cog) {
        const int co = g * a.Cog + co_g;
        float e;
        if (INJ) {
          e = a.eps_b[(long long)s * a.Co + co];
        } else {
          RngKey kb = key_w;
          kb.layer_tensor = layer_tensor_word(a.layer_id, 1);
          float z[4];
          philox_normal4(kb, sample, (uint32_t)(co >> 2), z);
          const int sel = co & 3;
          e = sel == 0 ? z[0] : sel == 1 ? z[1] : sel == 2 ? z[2] : z[3];
        }
        const float dl = __fmul_rn(softplus(a.rho_b[co]), e);
        b0 = FLIP ? a.mu_b[co] : __fadd_rn(a.mu_b[co], dl);
        b1 = dl;
      }
      bias0[ptid] = b0;
      if (FLIP) bias1[ptid] = b1;
      const bool cv = a.ep_scale && co_g < a.Cog;
      const int cs = cv ? g * a.Cog + co_g : 0;
      const float sc = a.ep_scale ? a.ep_scale[cs] : 1.f, sh = a.ep_shift ? a.ep_shift[cs] : 0.f;
      osc[ptid] = cv ? sc : 1.f;
      osh[ptid] = cv ? sh : 0.f;
    }
    __syncthreads();
    if (TRANS && a.out_vec4) {  // the consumers pass the output tile through LDS: same barriers ...
      if constexpr (kReadoutAll) {  // ... and a share of the read-out
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          if (i > 0) __syncthreads();
          __syncthreads();
          if constexpr (!POOL) readout_quads(i, tid);
          else if (pool_pow2) readout_pool(i, wave);
        }
      } else {
        for (int i = 0; i < 2 * TN - 1; ++i) __syncthreads();
      }
    }
  } else {
    // =================================================== CONSUMERS ===========================================================
    if (kl_block) {
      // KL: sweep this workgroup's slice of the weights while the producers fill stage 0, then publish one fp64 partial
      // PER WAVE right away -- before this workgroup has dirtied L2 with its outputs, so the agent-scope release that the
      // hand-off needs has almost nothing to write back -- and let the last-arriving wave finish in fixed slot order.
      long long chunk = (a.w_elems + a.kl_slices - 1) / a.kl_slices;
      chunk = (chunk + 3) & ~3ll;
      const long long lo = (long long)blockIdx.x * chunk;
      const long long hi = (lo + chunk < a.w_elems) ? lo + chunk : a.w_elems;
      const bool v4 = ((((uintptr_t)a.mu_w | (uintptr_t)a.rho_w | (uintptr_t)a.pmu_w | (uintptr_t)a.psig_w) & 15u) == 0);
      double kl_acc = 0.0;
      long long i = lo + 4ll * ptid;
      if (v4) {
        for (; i + 3 < hi; i += 1024) {
          const float4 m4 = *reinterpret_cast<const float4*>(a.mu_w + i), r4 = *reinterpret_cast<const float4*>(a.rho_w + i);
          const float4 p4 = *reinterpret_cast<const float4*>(a.pmu_w + i), q4 = *reinterpret_cast<const float4*>(a.psig_w + i);
          const float t0 = kl_term(m4.x, softplus(r4.x), p4.x, q4.x) + kl_term(m4.y, softplus(r4.y), p4.y, q4.y);
          const float t1 = kl_term(m4.z, softplus(r4.z), p4.z, q4.z) + kl_term(m4.w, softplus(r4.w), p4.w, q4.w);
          kl_acc += (double)t0 + (double)t1;
        }
      }
      for (; i < hi; i += 1024)  // tail quad / unaligned bases
        for (int j = 0; j < 4; ++j)
          if (i + j < hi) kl_acc += (double)kl_term(a.mu_w[i + j], softplus(a.rho_w[i + j]), a.pmu_w[i + j], a.psig_w[i + j]);
      const double wsum = wave_sum(kl_acc);
      const int nslots = 4 * a.kl_slices;
      int last = 0;
      if (lane == 0) last = publish_and_ticket_wt(a.slots, a.counter, (int)blockIdx.x * 4 + wave, wsum, (unsigned)nslots) ? 1 : 0;
      if (__builtin_amdgcn_readfirstlane(last)) {  // this wave arrived last: every slot is published
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
    // LDS column base of this lane's output pixel per 32-wide column group
    int colbase[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int ml = wm * WTM + j * 32 + li;
      if (LINEAR) {
        colbase[j] = ml;
      } else {
        int b, ho, wo;
        const bool live = col_decode(ml, b, ho, wo);
        colbase[j] = live ? (b - b0) * PIMG + (ho - r0) * ps_h * PWt + (wo - w0) * ps_w : 0;
      }
    }
    __syncthreads();  // stage 0 staged, rowtab written

    // K-row offsets of this lane half, in registers for the whole kernel (the stage shape never changes)
    int rowoff[kBK / 2];
#pragma unroll
    for (int q = 0; q < kBK / 2; ++q) rowoff[q] = (n_ach == 1 && 2 * q < KC) ? rowtab[2 * q + lh] : 0;
    const int wrow = lh * WS + wn * WTN + li;

    f32x16 acc[NW][TN][TM];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[w][i][j][r] = 0.f;

    const bool stamp = dbg_ && blockIdx.x == 0 && tid == 0;
    if (stamp) dbg_[0] = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < NS; ++st) {
      if (stamp && st < 60) dbg_[2 + 2 * st] = __builtin_amdgcn_s_memtime();
      const int cch = n_ach == 1 ? st : st / n_ach;
      int KCs = KC;  // K rows of this stage
      if (n_ach > 1) {
        const int ach = st - cch * n_ach;
        KCs = ((nA - ach * NA) < NA ? (nA - ach * NA) : NA) << lcc;
#pragma unroll
        for (int q = 0; q < kBK / 2; ++q) rowoff[q] = (2 * q < KCs) ? rowtab[(st & 1) * 40 + 2 * q + lh] : 0;
      }
      const float* const Wt0 = wbuf + (st & 1) * NW * W_WORDS + wrow;
      const float* const Wt1 = Wt0 + W_WORDS;
      const float* const Xt0 = xbuf + (cch & 1) * NW * X_WORDS;
      const float* const Xt1 = Xt0 + X_WORDS;
      float af[2][NW][TN], bf[2][NW][TM];
      auto load_frags = [&](auto slotc, auto qc) {
        constexpr int slot = decltype(slotc)::value, q = decltype(qc)::value;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          af[slot][0][i] = Wt0[2 * q * WS + i * 32];
          if (FLIP) af[slot][NW - 1][i] = Wt1[2 * q * WS + i * 32];
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          bf[slot][0][j] = Xt0[rowoff[q] + colbase[j]];
          if (FLIP) bf[slot][NW - 1][j] = Xt1[rowoff[q] + colbase[j]];
        }
      };
      auto mfmas = [&](auto slotc) {
        constexpr int slot = decltype(slotc)::value;
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
              acc[w][i][j] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x2f32(bf[slot][w][j], af[slot][w][i], acc[w][i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][w][i], bf[slot][w][j], acc[w][i][j], 0, 0, 0);
      };
      // 18 statically unrolled steps (2 K rows each), guarded by the stage's row count; fragments of step q+1 are
      // fetched while the MFMAs of step q run
      load_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      auto step_pair = [&](auto qc) {
        constexpr int q = decltype(qc)::value;  // even step index
        if (2 * q < KCs) {                      // uniform; the row count is a multiple of 4, so steps come in pairs
          load_frags(std::integral_constant<int, 1>{}, std::integral_constant<int, q + 1>{});
          mfmas(std::integral_constant<int, 0>{});
          if constexpr (q + 2 < kBK / 2) {
            if (2 * (q + 2) < KCs) load_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, q + 2>{});
          }
          mfmas(std::integral_constant<int, 1>{});
        }
      };
      step_pair(std::integral_constant<int, 0>{});
      step_pair(std::integral_constant<int, 2>{});
      step_pair(std::integral_constant<int, 4>{});
      step_pair(std::integral_constant<int, 6>{});
      step_pair(std::integral_constant<int, 8>{});
      step_pair(std::integral_constant<int, 10>{});
      step_pair(std::integral_constant<int, 12>{});
      step_pair(std::integral_constant<int, 14>{});
      step_pair(std::integral_constant<int, 16>{});
      if (stamp && st < 60) dbg_[2 + 2 * st + 1] = __builtin_amdgcn_s_memtime();
      __syncthreads();
    }
    if (stamp) dbg_[1] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // the producers have staged bias / output-stage constants

    // ---- output stage + store ----
    const float* const sout_s = (FLIP && INJ) ? a.sign_out + (long long)s * a.out_elems : nullptr;
    if (TRANS && a.out_vec4) {
      // Spatial NCHW output. The accumulators come out of the D[m][co] orientation with a lane owning ONE output channel
      // (bias / scale / shift are lane constants) and registers 4q..4q+3 holding 4 consecutive output positions; stored
      // from there, the 64 lanes of a wave would hit 64 different 128-byte lines with 16 bytes each. So the tile goes
      // through LDS, 32 * CWN channels at a time: [channel][BM positions], read back with consecutive lanes on
      // consecutive 16-byte pieces of one channel -> whole lines per store (and per residual load). The host guarantees
      // Wo % 4 == 0, tile widths that are multiples of 4 and 16-byte aligned tensors: a quad never leaves its row.
      constexpr int SROW = BM + 4, SROWS = 32 * CWN;
      static_assert(4 * BN + SROWS * SROW <= 2 * BUF_WORDS, "output staging fits the operand buffers");
      float* const stage = smem + 4 * BN;  // past the bias / scale / shift vectors
      float bsv[TN], b1v[TN], scv[TN], shv[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int co_l = wn * WTN + i * 32 + li;
        bsv[i] = bias0[co_l], scv[i] = osc[co_l], shv[i] = osh[co_l];
        b1v[i] = FLIP ? bias1[co_l] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if (i > 0) __syncthreads();  // the previous pass has been read out
        float* const srow = stage + (wn * 32 + li) * SROW + wm * WTM + 4 * lh;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[4];
            uint32_t oidx = 0;
            if constexpr (FLIP) {
              int bq, hq, wq;
              const bool mok = col_decode(wm * WTM + j * 32 + 8 * q + 4 * lh, bq, hq, wq);
              const int co_l = wn * WTN + i * 32 + li;
              oidx = (mok && n0 + co_l < a.Cog) ? (uint32_t)(((bq * a.Co + g * a.Cog + n0 + co_l) * a.Ho + hq) * a.Wo + wq) : 0u;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = __fadd_rn(acc[0][i][j][4 * q + e], bsv[i]);
              if constexpr (FLIP) {
                const float so = INJ ? sout_s[oidx + e] : hash_sign(skey_out, oidx + e);
                v[e] = __fadd_rn(v[e], __fmul_rn(__fadd_rn(acc[NW - 1][i][j][4 * q + e], b1v[i]), so));
              }
              v[e] = __fadd_rn(__fmul_rn(v[e], scv[i]), shv[i]);
            }
            *reinterpret_cast<float4*>(srow + j * 32 + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
        __syncthreads();
        if (stamp) dbg_[120 + 2 * i] = __builtin_amdgcn_s_memtime();
        if constexpr (POOL) {
          // fused MaxPool2d(3, 2, 1) of whole staged images (tile column = (img * Ho + ho) * Wo + wo); NaN wins like torch's
          const int Hp = a.ep_Hp, Wp = a.ep_Wp, PP = Hp * Wp, per_row = t_NI * PP;
          if (pool_pow2) {
            readout_pool(i, wave);
            if (stamp) dbg_[121 + 2 * i] = __builtin_amdgcn_s_memtime();
            continue;
          }
          const uint32_t inv_row = (uint32_t)((0x100000000ull + (unsigned)per_row - 1) / (unsigned)per_row);
          const uint32_t inv_pp = (uint32_t)((0x100000000ull + (unsigned)PP - 1) / (unsigned)PP);
          const uint32_t inv_wp = (uint32_t)((0x100000000ull + (unsigned)Wp - 1) / (unsigned)Wp);
          for (int c = tid; c < SROWS * per_row; c += 256) {  // < 2^16: the magic divisions are exact
            const int row = per_row == 1 ? c : (int)__umulhi((uint32_t)c, inv_row), rem = c - row * per_row;
            const int img = PP == 1 ? rem : (int)__umulhi((uint32_t)rem, inv_pp), pp = rem - img * PP;
            const int py = Wp == 1 ? pp : (int)__umulhi((uint32_t)pp, inv_wp), px = pp - py * Wp;
            const int co_l = (row >> 5) * WTN + i * 32 + (row & 31), b = b0 + img;
            if (b < a.B && n0 + co_l < a.Cog) {
              const float* const plane = stage + row * SROW + img * RW;
              // all nine loads first (clamped addresses, no branches): one LDS latency per output instead of nine
              float v[9];
#pragma unroll
              for (int dy = 0; dy < 3; ++dy) {
                const int y = 2 * py - 1 + dy;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                  const int x = 2 * px - 1 + dx;
                  const bool in = (unsigned)y < (unsigned)a.Ho && (unsigned)x < (unsigned)a.Wo;
                  const float t = plane[in ? y * a.Wo + x : 0];
                  v[dy * 3 + dx] = in ? t : -INFINITY;  // padding never wins (and is not NaN)
                }
              }
              float m = -INFINITY;
#pragma unroll
              for (int k = 0; k < 9; ++k) m = (v[k] > m || v[k] != v[k]) ? v[k] : m;
              if (relu) m = m < 0.f ? 0.f : m;  // max and ReLU commute
              out_s[((b * a.Co + g * a.Cog + n0 + co_l) * Hp + py) * Wp + px] = m;
            }
          }
          if (stamp) dbg_[121 + 2 * i] = __builtin_amdgcn_s_memtime();
          continue;
        }
        readout_quads(i, tid);
      }
    } else {
  #pragma unroll
      for (int j = 0; j < TM; ++j) {
        int b_col = 0, h_col = 0, w_col = 0;
        bool live_col = false;
        if (!TRANS) live_col = col_decode(wm * WTM + j * 32 + li, b_col, h_col, w_col);  // lanes run along the tile columns
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          uint32_t oi[16];
          bool okv[16];
          int col[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            int co_l;
            bool live;
            int bb = b_col, hh = h_col, ww = w_col;
            if (TRANS) {
              co_l = wn * WTN + i * 32 + li;
              live = col_decode(wm * WTM + j * 32 + row, bb, hh, ww);
            } else {
              co_l = wn * WTN + i * 32 + row;
              live = live_col;
            }
            okv[r] = live && n0 + co_l < a.Cog;
            col[r] = co_l;
            oi[r] = okv[r] ? (uint32_t)(((bb * a.Co + g * a.Cog + n0 + co_l) * a.Ho + hh) * a.Wo + ww) : 0u;
          }
          float rsd[16], so[FLIP ? 16 : 1];
          if (res_s) {
  #pragma unroll
            for (int r = 0; r < 16; ++r) rsd[r] = res_s[oi[r]];
          } else {
  #pragma unroll
            for (int r = 0; r < 16; ++r) rsd[r] = 0.f;
          }
          if constexpr (FLIP) {
  #pragma unroll
            for (int r = 0; r < 16; ++r) so[r] = INJ ? sout_s[oi[r]] : hash_sign(skey_out, oi[r]);
          }
  #pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = __fadd_rn(acc[0][i][j][r], bias0[col[r]]);
            if constexpr (FLIP) v = __fadd_rn(v, __fmul_rn(__fadd_rn(acc[NW - 1][i][j][r], bias1[col[r]]), so[r]));
            v = __fadd_rn(__fmul_rn(v, osc[col[r]]), osh[col[r]]);
            v = __fadd_rn(v, rsd[r]);
            v = (relu && v < 0.f) ? 0.f : v;
            if (okv[r]) out_s[oi[r]] = v;
          }
        }
      }
    }
  }

  if (dbg_ && blockIdx.x == 0 && tid == 0) dbg_[126] = __builtin_amdgcn_s_memtime();
}

}  // namespace bt
