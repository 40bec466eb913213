// Split-precision flavour of the fused forward: the fp32 contraction as exact bf16 pieces on the bf16 matrix pipe.
//
// gfx950 runs fp32-input MFMA at the VECTOR rate (64 FLOP/clk/SIMD) and an fp32 MFMA blocks the SIMD's VALU issue, so
// the fp32 kernels (bt_fused_fast.h) are bound by t_MFMA + t_VALU. v_mfma_f32_32x32x16_bf16 has 16x the rate and holds
// the VALU issue for only 8 of its 32 cycles. Here every fp32 operand value v is cut into NP bf16 pieces,
//     v = h + m (+ l),   h = top 16 bits of v,   m = top 16 bits of (v - h),   l = v - h - m
// (each difference is exact in fp32, and with NP = 3 the three pieces hold all 24 significant bits: the split is
// EXACT), and a product x*w becomes the sum of the piece products of weight >= 2^-16:
//     NP = 3, 6 terms:  xh*wh + xh*wm + xm*wh + xh*wl + xm*wm + xl*wh        (dropped terms <= 2^-24 |x*w|: fp32 level)
//     NP = 2, 3 terms:  xh*wh + xh*wm + xm*wh                                   (dropped terms <= 2^-15 |x*w|: opt-in)
// Each piece product is exact in fp32 (8 x 8 significant bits) and the MFMA accumulates in fp32, so the 6-term form
// has the accuracy of the fp32 FMA chain at 6/16 of its matrix-pipe time, with the weight synthesis co-issuing.
//
// Same algorithm, same draw stream (tap-major Philox blocks, one block = 4 channels of one tap), same KL sweep and
// output stage as bt_fused_fast.h; what changes is the K order and the LDS images.
//   * K order (canonical, independent of the tile): channels in OCTETS of 8; an MFMA step covers two (octet, active tap)
//     entries -- lanes 0-31 hold the 8 channels of the first, lanes 32-63 of the second. nA > 1: the octet's taps in
//     pairs (a0,a1), (a2,a3) ...; an odd last tap runs with an empty second half. nA == 1: consecutive octets in pairs.
//   * x tile: the LDS patch of the fast kernel, but channel-last and WITHOUT its zero halo -- X[octet][real patch pixel][piece]
//     [8 channels] bf16, 16*NP bytes per pixel, so an operand fragment (8 channels of one pixel, one piece) is ONE
//     ds_read_b128 and the three pieces are immediates apart. The byte address of every (column group, step) fragment of a
//     lane is computed once per workgroup; a (lane, tap) pair that falls into the padding points at one shared zero pixel
//     (same address in many lanes = a broadcast). 8x8 maps keep 64 instead of 100 pixels per image, 4x4 maps 16 of 36.
//   * w tile: W[step][lane half][piece][row][8 channels] bf16 -- 16 consecutive rows cover all 64 banks.
// Producers split every sampled weight and every staged activation once; consumers only read LDS and issue MFMAs.
#pragma once
#include "bt_fused_fwd.h"

namespace bt {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kSplitSteps = 5;  // MFMA K16-steps per barrier stage (9 taps of one octet = 5 steps)

// FLIP (Flipout): two weight images per stage (mu and sigma*eps) and a fourth 16-byte piece per x pixel (the sign masks of
// its 8 channels), which leaves room for 301 patch pixels.
template <int BM, bool FLIP = false>
constexpr int split_xpo() {  // pixel-octet slots of one x buffer: 1024 patch pixels (what 160 KB allow) + the shared zero pixel
  return FLIP ? 302 : 1025;
}
template <int BN, int NP, bool FLIP = false>
constexpr int split_w_bytes() { return (FLIP ? 2 : 1) * kSplitSteps * 2 * NP * BN * 16; }
template <int BM, int NP, bool FLIP = false>
constexpr int split_x_bytes() { return split_xpo<BM, FLIP>() * (NP + (FLIP ? 1 : 0)) * 16; }
constexpr int kSplitMiscBytes = kMaxTaps * 16 + 96 + 32 + 64;
template <int BN, int BM, int NP, bool FLIP = false>
constexpr int split_lds_bytes() { return 2 * (split_w_bytes<BN, NP, FLIP>() + split_x_bytes<BM, NP, FLIP>()) + kSplitMiscBytes; }

// fp32 -> bf16 pieces by truncation, as fp32 bit patterns whose upper halves are the pieces
__device__ __forceinline__ void split_pieces(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(v) & 0xFFFF0000u;
  const float r = __fsub_rn(v, __uint_as_float(h));  // exact
  m = __float_as_uint(r) & 0xFFFF0000u;
  l = __float_as_uint(__fsub_rn(r, __uint_as_float(m)));  // exact; <= 8 significant bits: its lower half is zero
}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t hi, uint32_t lo) {  // (hi.upper16 << 16) | lo.upper16
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
// The same split for two values at once, straight to the packed bf16 pairs (b in the upper half): the two exact subtractions run as
// packed fp32 operations (v_pk_add_f32: one instruction for both values) -- 7 instead of 9.5 instructions per pair. Same IEEE
// operations, so the pieces are bit for bit split_pieces'.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  const f32x2_t v = {a, b};
  const u32x2_t hb = __builtin_bit_cast(u32x2_t, v) & 0xFFFF0000u;
  const f32x2_t r = v - __builtin_bit_cast(f32x2_t, hb);        // exact
  const u32x2_t mb = __builtin_bit_cast(u32x2_t, r) & 0xFFFF0000u;
  const f32x2_t q = r - __builtin_bit_cast(f32x2_t, mb);        // exact; <= 8 significant bits
  const u32x2_t lb = __builtin_bit_cast(u32x2_t, q);
  h = pack_hi16(hb.y, hb.x), m = pack_hi16(mb.y, mb.x), l = pack_hi16(lb.y, lb.x);
}

// n / d with the host's reciprocal (bt_fused_split_host.h: inv == 0 -> divide; d == 1 needs none)
__device__ __forceinline__ int udiv_inv(int n, int d, uint32_t inv) {
  return inv ? (int)__umulhi((uint32_t)n, inv) : (d == 1 ? n : n / d);
}
// ceil(2^32 / d), d >= 2, through ONE 32-bit division: floor((2^32 - 1) / d) + 1
__device__ __forceinline__ uint32_t inv32(int d) { return 0xFFFFFFFFu / (uint32_t)d + 1u; }
__device__ __forceinline__ int div_small(int n, int d) { return d == 1 ? n : d == 2 ? n >> 1 : n / d; }  // n >= 0; strides are 1 or 2 in practice

// NP: pieces per value (3: exact split, 6 product terms; 2: 3 terms). 4 consumer waves (64 x BM/4 each) + NPW producer waves
// (8 on the 128-wide tile of the small feature maps, where one draw serves few columns and the accumulators are small).
// XM: how the x patch is fetched. A producer thread owns ITEMS f = ptid + kProducers i of the stage's NO octet planes (the same
// ones in every stage, so their addresses are decoded once) and issues all their loads one stage ahead.
//   0: item = (patch pixel, octet): one dword per channel; any geometry.
//   1: 1x1 input planes (and Linear): item = (image, octet), its 8 channels are 32 contiguous bytes.
//   2: 2x2 input planes wholly inside the patch: item = (image, channel pair) -- a channel's plane is 16 contiguous bytes and
//      consecutive lanes take consecutive channels, so a wave's loads cover whole 256-byte pieces of an image.
//   3: tiles of whole rows with W % 4 == 0 whose patch keeps every column of its rows (stride 1, or a strided 3x3 -- the patch
//      of a strided window is the stride-1 grid): item = (4 consecutive input pixels, half an octet): 16-byte row pieces.
//   4: the same for a patch that keeps every SECOND column (1x1 / stride 2: the downsamples): item = (4 consecutive input
//      pixels, octet) = 2 patch pixels x 8 channels. (A dword gather at stride 2 costs the texture path ~70 cycles per wave
//      instruction against 16 for a contiguous 16-byte one: 9.4 K -> 2 K cycles per stage on ResNet18's downsamples.)
//
// FLIP: the Flipout forward  out = x*mu + s_out o ((x o s_in) * (sigma o eps))  (flipout_layers.py:conv / linear forward): two
// contractions that share the x pieces. The producers stage mu and sigma*eps as two weight images and, next to the pieces of
// every x pixel, the sign masks of its 8 channels (bit 15 of each bf16 lane; same hash stream as the fp32 kernels); a consumer
// issues the 6 terms of x*mu, flips the sign bits of its x fragments in registers (-(h+m+l) = -h-m-l: the split of -v is the
// negated split of v), and issues the 6 terms of (x o s_in)*(sigma*eps) into the second accumulator set. 4 consumer waves of
// 32 channels x BM/2 pixels each (2 x 2), BM = 256 or 128. s_out meets the second set in the read-out.
template <int BN, int BM, int NP, int NPW, int XM, bool FLIP = false>
__global__ __launch_bounds__(256 + 64 * NPW) void fused_split_kernel(const FwdArgs a) {
  static_assert((BN == 64 && (BM == 512 || BM == 256 || BM == 128)) || (BN == 32 && BM == 128 && !FLIP), "tile shapes of this flavour");
  static_assert(!FLIP || ((BM == 256 || BM == 128) && NP == 3), "Flipout: the 64 x 256 / 64 x 128 tiles, exact split");
  constexpr int kProducers = 64 * NPW, kThreadsAll = 256 + kProducers;
  constexpr int CWM = FLIP ? 2 : 4, CWN = 4 / CWM, WTM = BM / CWM, TN = BN / CWN / 32, TM = WTM / 32;
  constexpr int NOP = FLIP ? 2 : 1;            // weight operands (images per stage) = accumulator sets
  constexpr int PB = 16 * (NP + (FLIP ? 1 : 0));  // bytes per (pixel, octet) of the x patch
  constexpr int W_BYTES = split_w_bytes<BN, NP, FLIP>(), W_OP = W_BYTES / NOP, X_BYTES = split_x_bytes<BM, NP, FLIP>(), XPO = split_xpo<BM, FLIP>();
  constexpr int W_STEP = 2 * NP * BN * 16, W_HALF = NP * BN * 16, W_PIECE = BN * 16;
  // output staging: all BN channels in one pass when the operand buffers are large enough, else 32 at a time
  constexpr int SROWS = ((4 * BN + NOP * BN * (BM + 4)) * 4 <= 2 * (W_BYTES + X_BYTES)) ? BN : 32, NPASS = BN / SROWS;
  static_assert(!FLIP || NPASS == 1, "Flipout stages both accumulator sets in one pass");

  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  char* const wbuf = smem_c;                  // [2][W_BYTES]
  char* const xbuf = smem_c + 2 * W_BYTES;    // [2][X_BYTES]
  float* const smem = reinterpret_cast<float*>(smem_c);
  int4* const taptab = reinterpret_cast<int4*>(smem_c + 2 * (W_BYTES + X_BYTES));

  unsigned long long* const dbg_ = kStamps ? a.dbg : nullptr;  // stage stamps: diagnostic build only (make STAMPS=1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (dbg_ && dbg_[201] && tid == 0) {  // diagnostic build: per-workgroup timeline (100 MHz wall clock, shader clock, HW_ID)
    dbg_[256 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    dbg_[256 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
    dbg_[256 + 4 * blockIdx.x + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                                     ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);  // HW_ID, XCC_ID
  }
  const bool stamp0 = dbg_ && blockIdx.x == 0 && (tid == 0 || tid == 256);
  if (stamp0 && tid == 0) dbg_[210] = __builtin_amdgcn_s_memtime();
  const bool producer = wave >= 4;
  const int ptid = producer ? tid - 256 : tid;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & (CWM - 1), wn = (wave & 3) / CWM;  // consumer wave: column block, row block

  int L = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, a.total_blocks));
  int Lq = udiv_inv(L, a.m_tiles, a.inv_m_tiles);
  const int mt = __builtin_amdgcn_readfirstlane(L - Lq * a.m_tiles);
  L = Lq, Lq = udiv_inv(L, a.S, a.inv_S);
  const int s = __builtin_amdgcn_readfirstlane(L - Lq * a.S);
  L = Lq, Lq = udiv_inv(L, a.n_tiles, a.inv_n_tiles);
  const int nt = __builtin_amdgcn_readfirstlane(L - Lq * a.n_tiles);
  const int g = __builtin_amdgcn_readfirstlane(Lq);
  const int n0 = nt * BN;
  const bool pix = a.pixel_major != 0;
  const int t_NI = a.t_NI, t_R = a.t_R, t_Wt = a.t_Wt, RW = t_R * t_Wt, Mt = t_NI * RW;
  const int trest = udiv_inv(mt, a.n_bt, a.inv_n_bt);
  const int bt = __builtin_amdgcn_readfirstlane(mt - trest * a.n_bt);
  const int rt = __builtin_amdgcn_readfirstlane(udiv_inv(trest, a.n_ct, a.inv_n_ct)), ct = __builtin_amdgcn_readfirstlane(trest - rt * a.n_ct);
  const int b0 = bt * t_NI, r0 = rt * t_R, w0 = ct * t_Wt;
  const uint32_t inv_rw = RW > 1 ? (a.inv_rw ? a.inv_rw : inv32(RW)) : 0u;
  const uint32_t inv_wt = t_Wt > 1 ? (a.inv_wt ? a.inv_wt : inv32(t_Wt)) : 0u;
  auto col_decode = [&](int ml, int& b, int& ho, int& wo) -> bool {  // tile column -> output coordinates; false: dead column
    const int img = RW == 1 ? ml : (int)__umulhi((uint32_t)ml, inv_rw);
    const int rem = ml - img * RW;
    const int r = t_Wt == 1 ? rem : (int)__umulhi((uint32_t)rem, inv_wt);
    b = b0 + img, ho = r0 + r, wo = w0 + (rem - r * t_Wt);
    return ml < Mt && b < a.B && ho < a.Ho && wo < a.Wo;
  };
  const uint32_t sample = a.sample0 + (uint32_t)s;
  const int T = a.T, Cig = a.Cig;

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

  // ---- active taps of this tile + their window. Every wave computes them (identical values: the table in LDS is written by
  // all waves alike, and a wave reads it behind its own writes), so no workgroup barrier opens the kernel. -------------------
  int nA, dymin, dymax, dxmin, dxmax;
  {
    bool act = false;
    int4 e = make_int4(0, 0, 0, 0);
    if (lane < T) {
      const int kh = udiv_inv(lane, a.KW, a.inv_kw), kw = lane - kh * a.KW;
      e = make_int4(0, kh * a.DH, kw * a.DW, lane);
      if (pix) {
        act = (unsigned)(r0 * a.SH - a.PH + e.y) < (unsigned)a.H && (unsigned)(w0 * a.SW - a.PW + e.z) < (unsigned)a.W;
      } else {
        const int lo_h = a.PH - e.y, lo_w = a.PW - e.z;
        const int hc = lo_h > 0 ? div_small(lo_h + a.SH - 1, a.SH) : 0, wc = lo_w > 0 ? div_small(lo_w + a.SW - 1, a.SW) : 0;
        act = hc < a.Ho && hc * a.SH - lo_h < a.H && wc < a.Wo && wc * a.SW - lo_w < a.W;
        if (a.row_taps) {  // tiles of ONE output row (images x row r0): only the kernel rows that meet data for THIS row
          act = act && (unsigned)(r0 * a.SH - a.PH + e.y) < (unsigned)a.H;
        }
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
    nA = __builtin_amdgcn_readfirstlane((int)__popcll(mask));  // 0: degenerate geometry, outputs are the bias alone
    dymin = __builtin_amdgcn_readfirstlane(dy0), dymax = __builtin_amdgcn_readfirstlane(dy1);
    dxmin = __builtin_amdgcn_readfirstlane(dx0), dxmax = __builtin_amdgcn_readfirstlane(dx1);
  }
  if (stamp0 && tid == 0) dbg_[211] = __builtin_amdgcn_s_memtime();

  // ---- stage shape ------------------------------------------------------------------------------------------------------
  // nA > 1: SPO = ceil(nA/2) steps per octet, NO octets per stage. nA == 1: one step per PAIR of octets.
  const int G8 = Cig >> 3;  // host: Cig % 8 == 0
  const int ps_h = (dymax == dymin) ? 1 : a.SH, gs_h = (dymax == dymin) ? a.SH : 1;
  const int ps_w = (dxmax == dxmin) ? 1 : a.SW, gs_w = (dxmax == dxmin) ? a.SW : 1;
  const int PHt = (t_R - 1) * ps_h + (dymax - dymin) + 1, PWt = (t_Wt - 1) * ps_w + (dxmax - dxmin) + 1;
  const int x_lo = w0 * a.SW - a.PW + dxmin, y_lo = r0 * a.SH - a.PH + dymin;
  // Only the patch rows / columns that exist in the image are stored: grid rows k (input row y_lo + k * gs_h) with
  // kmin_h <= k <= kmax_h, likewise columns. (PHt x PWt is the full window, halo included.)
  const int kmin_h = y_lo < 0 ? div_small(-y_lo + gs_h - 1, gs_h) : 0, kmin_w = x_lo < 0 ? div_small(-x_lo + gs_w - 1, gs_w) : 0;
  int kmax_h = a.H - 1 - y_lo >= 0 ? div_small(a.H - 1 - y_lo, gs_h) : -1, kmax_w = a.W - 1 - x_lo >= 0 ? div_small(a.W - 1 - x_lo, gs_w) : -1;
  kmax_h = kmax_h < PHt - 1 ? kmax_h : PHt - 1, kmax_w = kmax_w < PWt - 1 ? kmax_w : PWt - 1;
  const int NYR = kmax_h >= kmin_h ? kmax_h - kmin_h + 1 : 0, NXR = kmax_w >= kmin_w ? kmax_w - kmin_w + 1 : 0;
  const int PIMG = NYR * NXR, PCH = t_NI * PIMG;  // real patch pixels of one image / of one octet plane (host: NO * PCH < XPO)
  constexpr int ZOFF = X_BYTES - PB;                // the shared zero pixel: last slot of each x buffer, never written
  const int SPO = (nA + 1) >> 1;
  int NO;
  if (nA <= 1) {
    NO = 2 * kSplitSteps;
  } else {
    NO = SPO == 1 ? 5 : SPO == 2 ? 2 : 1;  // floor(kSplitSteps / SPO), SPO <= 5 (host: at most 9 taps)
    static_assert(kSplitSteps == 5, "the table above");
  }
  while (NO > 1 && NO * PCH > XPO - 1) --NO;
  if (nA <= 1 && NO > 1) NO &= ~1;  // whole pairs per stage
  if (NO > G8) NO = G8;             // (a single stage may then end in a half-empty pair: the W slot holds zeros)
  const int NQ = NO * (nA > 0 ? nA : 1);                       // (octet, tap) entries of a full stage
  const int NSTEP = nA <= 1 ? (NO + 1) >> 1 : NO * SPO;        // MFMA steps of a full stage
  const int NS = nA ? (G8 + NO - 1) / NO : 0;
  // The W slots no unit ever writes (second half of an odd tap count's last step) must hold zeros: clear the W buffers once.
  // Every x slot a consumer reads is either a real patch pixel of an octet plane of the stage -- written in every stage, with
  // zeros past the end of the batch / of the channels -- or the shared zero pixel (padding taps, dead columns, the missing
  // half of an odd last pair): clear the two zero pixels. (The barrier that publishes the clear sits in both arms below,
  // behind the work that needs no LDS: the producers have issued stage 0's loads by then.)
  for (int i = tid; i < 2 * W_BYTES / 16; i += kThreadsAll) reinterpret_cast<uint4*>(smem_c)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 2 * (PB / 16)) reinterpret_cast<uint4*>(xbuf + (tid / (PB / 16)) * X_BYTES + X_BYTES - PB)[tid % (PB / 16)] = make_uint4(0, 0, 0, 0);

  const float* const xs = a.x + (long long)s * a.x_sample_stride;
  constexpr uint32_t kOOB = 0x80000000u;
  // Offset of a guarded load: `off` when the element exists, else an out-of-range offset (the buffer load returns 0). As bit
  // arithmetic on purpose: written as a select, the compiler sinks the offset computation into a branch per load, and the
  // wait-count pass then drains the loads in flight at those block boundaries whenever two of them share a register
  // (measured: 2.5x on the short-stage 1x1 layers of ResNet50 after an unrelated change moved the register allocation).
  auto guard_off = [](bool in, uint32_t off) -> uint32_t {
    const uint32_t m = 0u - (uint32_t)in;
    return (off & m) | (0x80000000u & ~m);
  };
  const int Cig4 = Cig;  // a multiple of 8 here
  const int pk_bytes = a.Co * T * Cig4 * 4;
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
  if (stamp0 && tid == 0) dbg_[212] = __builtin_amdgcn_s_memtime();

  // Read-out of one staged pass of the output tile (SROWS channels) by every wave (see the output stage below). A thread's
  // items share one pixel quad (the thread count is a multiple of the quads per row) and step RSTEP channels: the quad is
  // decoded once, an item costs an add (this loop is bound by the VALU issue of its index arithmetic, not by the stores).
  auto readout_quads = [&](int pass, int t0) {
    constexpr int SROW = BM + 4, QROW = BM / 4, NT_ = kThreadsAll;
    static_assert(NT_ % QROW == 0, "a thread keeps its pixel quad");
    constexpr int RSTEP = NT_ / QROW, NITc = (SROWS + RSTEP - 1) / RSTEP, U = NITc < 8 ? NITc : 8;
    const float* const stage = smem + 4 * BN;
    const int m4 = t0 % QROW, row0 = t0 / QROW;
    int bq, hq, wq;
    const bool mok = col_decode(4 * m4, bq, hq, wq);
    const int HoWo_ = a.Ho * a.Wo;
    const int co0 = pass * SROWS + row0;  // first channel (inside the tile) of this thread
    const uint32_t obase = mok ? (uint32_t)(((bq * a.Co + g * a.Cog + n0 + co0) * a.Ho + hq) * a.Wo + wq) : 0u;
    const int rows_ok = a.Cog - n0 - pass * SROWS < SROWS ? a.Cog - n0 - pass * SROWS : SROWS;  // staged rows that are channels
    const float* const sp = stage + row0 * SROW + 4 * m4;
#pragma unroll
    for (int k0 = 0; k0 < NITc; k0 += U) {
      uint32_t oidx[U];
      bool okq[U];
      float4 v[U], r4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u, row = row0 + k * RSTEP;
        okq[u] = k < NITc && mok && row < rows_ok;
        oidx[u] = okq[u] ? obase + (uint32_t)(k * RSTEP * HoWo_) : 0u;
        const int rr = row < SROWS ? k * RSTEP : 0;   // (rows past the staged block: any staged address)
        v[u] = *reinterpret_cast<const float4*>(sp + rr * SROW);
        if constexpr (FLIP) {  // second accumulator set x s_out, then the output-stage constants (the fp32 kernels' order)
          const float4 d = *reinterpret_cast<const float4*>(sp + (SROWS + rr) * SROW);
          const int co_l = okq[u] ? co0 + k * RSTEP : 0;
          const float sc = osc[co_l], sh = osh[co_l];
          v[u].x = __fadd_rn(__fmul_rn(__fadd_rn(v[u].x, __fmul_rn(d.x, hash_sign(skey_out, oidx[u]))), sc), sh);
          v[u].y = __fadd_rn(__fmul_rn(__fadd_rn(v[u].y, __fmul_rn(d.y, hash_sign(skey_out, oidx[u] + 1u))), sc), sh);
          v[u].z = __fadd_rn(__fmul_rn(__fadd_rn(v[u].z, __fmul_rn(d.z, hash_sign(skey_out, oidx[u] + 2u))), sc), sh);
          v[u].w = __fadd_rn(__fmul_rn(__fadd_rn(v[u].w, __fmul_rn(d.w, hash_sign(skey_out, oidx[u] + 3u))), sc), sh);
        }
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

  // The same staged tile read out by COLUMNS: maps whose rows are not 16-byte multiples (7x7, 14x14 rows of 7 / 14 pixels ...).
  // A thread keeps one tile column (the thread count is a multiple of BM) and walks the channels: consecutive lanes store
  // consecutive pixels of an image -- contiguous 4-byte stores instead of one scattered store per (lane = channel, pixel).
  const bool stage_cols = !a.out_vec4 && !pix && !a.row_taps && a.Ho * a.Wo > 4;
  auto readout_cols = [&](int pass, int t0) {
    constexpr int SROW = BM + 4, NT_ = kThreadsAll;
    static_assert(NT_ % BM == 0, "a thread keeps its column");
    constexpr int RSTEP = NT_ / BM, NITc = (SROWS + RSTEP - 1) / RSTEP;
    const float* const stage = smem + 4 * BN;
    const int c = t0 % BM, row0 = t0 / BM;
    int bq, hq, wq;
    const bool mok = col_decode(c, bq, hq, wq);
    const int HoWo_ = a.Ho * a.Wo;
    const int co0 = pass * SROWS + row0;
    const uint32_t obase = mok ? (uint32_t)(((bq * a.Co + g * a.Cog + n0 + co0) * a.Ho + hq) * a.Wo + wq) : 0u;
    const int rows_ok = a.Cog - n0 - pass * SROWS < SROWS ? a.Cog - n0 - pass * SROWS : SROWS;
    const float* const sp = stage + row0 * SROW + c;
#pragma unroll 8
    for (int k = 0; k < NITc; ++k) {
      const int row = row0 + k * RSTEP;
      const bool ok = mok && row < rows_ok;
      const uint32_t oi = ok ? obase + (uint32_t)(k * RSTEP * HoWo_) : 0u;
      const int rr = row < SROWS ? k * RSTEP : 0;
      float v = sp[rr * SROW];
      if constexpr (FLIP) {
        const float d = sp[(SROWS + rr) * SROW];
        const int co_l = ok ? co0 + k * RSTEP : 0;
        v = __fadd_rn(__fmul_rn(__fadd_rn(v, __fmul_rn(d, hash_sign(skey_out, oi))), osc[co_l]), osh[co_l]);
      }
      if (res_s && ok) v = __fadd_rn(v, res_s[oi]);
      v = (relu && v < 0.f) ? 0.f : v;
      if (ok) out_s[oi] = v;
    }
  };

  const int prio_mode = dbg_ ? (int)dbg_[200] : 0;  // diagnostic build: 0 producers first (product), 1 none, 2 consumers first
  if (producer) {
    // Producers first on the 512-wide tiles (layer1: 131.5 us against 139.0 without, tools/stamps.py --noprio) and in the Flipout flavour;
    // NOT on the narrower Reparameterization tiles, whose consumer stage is the longer one (profiles/r03_stamps_layer3.txt): same-box
    // A/B inside the cfg3 graph (tools/trace_layers.py, BT_LIB_PATH): the strided 3x3 layers 88.2 -> 84.8, 82.2 -> 79.1, 72.3 -> 66.5 us,
    // layer4's one-tap layers -2 us each, the 512-wide launches unchanged: 1928.7 -> 1909.1 us per step.
    if (prio_mode == 0 && (BM == 512 || FLIP)) __builtin_amdgcn_s_setprio(3);
    // =================================================== PRODUCERS ===========================================================
    // Weight unit u = (channel quad cq of the octet, row n, entry q = (octet ol, tap slot ai)): 4 sampled weights = one Philox
    // block. u = ptid + 256 i -> cq = u & 1, n = (u >> 1) % BN, q = (u >> 1) / BN: consecutive lanes fill one 16-byte LDS slot
    // in pairs and consecutive rows, so the ds_write_b64 of a wave touch every bank once.
    constexpr int UMAX = (2 * BN * 2 * kSplitSteps + kProducers - 1) / kProducers;
    const int nunits = 2 * BN * NQ;
    uint32_t e_off[UMAX];  // draw index (== element offset in the packed tensors) of the unit at octet 0 of stage 0
    int l_off[UMAX];       // LDS byte offset inside a W buffer; -1: no slot
    int u_ol[UMAX];        // octet inside the stage
    bool u_live[UMAX];     // wave-uniform: some lane of this wave holds a row that exists. A wave whose rows all lie past Cog (the
                           // classifier head: 10 of 64 rows) skips the unit -- loads, draws, split, LDS writes; its W slots keep
                           // the zeros of the initial clear.
    const uint32_t inv_na = nA > 1 ? inv32(nA) : 0u;
#pragma unroll
    for (int i = 0; i < UMAX; ++i) {
      const int u = ptid + kProducers * i;
      const int uu = u < nunits ? u : 0;
      int cq = uu & 1, n = (uu >> 1) & (BN - 1), q = (uu >> 1) / BN;
      if (nA == 1 && NO > 1) {  // one tap: a row's octets are contiguous in the packed tensors -> (quad, octet) fastest
        const uint32_t inv_no = inv32(NO);
        const int t2 = uu >> 1;
        n = (int)__umulhi((uint32_t)t2, inv_no);
        q = t2 - n * NO;
      }
      const int ol = nA > 1 ? (int)__umulhi((uint32_t)q, inv_na) : q;
      const int ai = nA > 1 ? q - ol * nA : 0;
      const int tap = nA ? taptab[ai].w : 0;
      const int st_ = nA > 1 ? ol * SPO + (ai >> 1) : (ol >> 1), hf = nA > 1 ? (ai & 1) : (ol & 1);
      const int co_g = n0 + n;
      const bool rv = co_g < a.Cog;
      const uint32_t co = (uint32_t)(g * a.Cog + (rv ? co_g : 0));
      e_off[i] = rv ? (co * (uint32_t)T + (uint32_t)tap) * (uint32_t)Cig4 + (uint32_t)(8 * ol + 4 * cq) : (kOOB >> 2);
      // (row slots are XOR-swizzled with the plane index: the units of one row that land in different (step, half) planes --
      //  consecutive lanes when one tap is active -- then write different banks; reads stay 16 distinct slots per lane group)
      l_off[i] = u < nunits ? st_ * W_STEP + hf * W_HALF + (n ^ ((2 * st_ + hf) & 7)) * 16 + cq * 8 : -1;
      u_ol[i] = ol;
      u_live[i] = __builtin_amdgcn_readfirstlane(__ballot(rv && u < nunits) != 0ull ? 1 : 0) != 0;   // (an SGPR condition: the guarded blocks below must stay UNIFORM branches -- as exec-masked regions they made the compiler drain every outstanding load at their entry, 2.5x on the short-stage layers)
    }
    const int wave_u0 = __builtin_amdgcn_readfirstlane(ptid & ~63);
    if (stamp0) dbg_[213] = __builtin_amdgcn_s_memtime();
    float4 mu[UMAX], rs[UMAX];
    auto load_w = [&](int st) {
      const int oct0 = st * NO;
#pragma unroll
      for (int i = 0; i < UMAX; ++i) {
        if (u_live[i]) {  // wave-uniform
          const bool in = l_off[i] >= 0 && oct0 + u_ol[i] < G8 && e_off[i] != (kOOB >> 2);
          const uint32_t sb = guard_off(in, 4u * (e_off[i] + (uint32_t)(8 * oct0)));
          mu[i] = ldf4(r_mu, sb), rs[i] = ldf4(r_rs, sb);
        }
      }
    };
    if (NS > 0) load_w(0);  // stage 0's weight loads fly while the x items are decoded
    // x items of this thread (see XM above): global byte offset of the item at octet 0, LDS byte offset inside an x buffer,
    // octet inside the stage.
    constexpr int PIT = (XM == 3 || XM == 4) ? ((XPO - 1) / 2 + kProducers - 1) / kProducers : (XPO - 1 + kProducers - 1) / kProducers;   // (XPO - 1 patch slots + the zero pixel)
    constexpr int XV = XM == 3 ? 16 : XM == 4 ? 32 : 8;
    int it_off[PIT], it_lds[PIT], it_ol[PIT];
    // real input rows of the patch and 16-byte quads per row (XM 3)
    // (XM 3: the stored columns are the whole row, NXR == W -- or, x_flat, the whole plane taken as ONE row of H*W pixels; XM 4: every second column)
    const bool flat = XM == 3 && a.x_flat != 0;
    const int ylo_r = y_lo + kmin_h * gs_h, nyr = flat ? 1 : NYR, W4 = flat ? a.HW >> 2 : a.W >> 2;
    const int xpar = (x_lo + kmin_w * gs_w) & 1;                          // XM 4: parity of the stored columns
    const int per_oct = XM == 2 ? 4 * t_NI : XM == 3 ? 2 * t_NI * nyr * W4 : XM == 4 ? t_NI * nyr * W4 : PCH;   // XM 2: (image, channel pair) items
    const int n_items = NO * per_oct;
    {
      const uint32_t inv_per = per_oct > 1 ? inv32(per_oct) : 0u;
      const uint32_t inv_pimg = PIMG > 0 ? inv32(PIMG) : 0u;
      const uint32_t inv_pw = NXR > 0 ? inv32(NXR) : 0u;
      const int qpi = nyr * W4;  // quads per image (XM 3)
      const uint32_t inv_qpi = qpi > 1 ? inv32(qpi) : 0u;
      const uint32_t inv_w4 = W4 > 1 ? inv32(W4) : 0u;
#pragma unroll
      for (int i = 0; i < PIT; ++i) {
        const int f = ptid + kProducers * i;
        const int ff = f < n_items ? f : 0;
        int ol = per_oct <= 1 ? ff : (int)__umulhi((uint32_t)ff, inv_per);
        int r = ff - ol * per_oct;
        if constexpr (XM == 1) {  // octet fastest: consecutive lanes read consecutive 32-byte pieces of one image
          const uint32_t inv_no = NO > 1 ? inv32(NO) : 0u;
          r = NO <= 1 ? ff : (int)__umulhi((uint32_t)ff, inv_no);
          ol = ff - r * NO;
        }
        int pair = 0;
        if constexpr (XM == 2) {  // channel pair fastest, then octet, then image: r = image
          const int cps = 4 * NO;   // channel pairs of a stage
          const uint32_t inv_cps = inv32(cps);
          r = (int)__umulhi((uint32_t)ff, inv_cps);
          const int cp = ff - r * cps;
          ol = cp >> 2, pair = cp & 3;
        }
        int off = (int)kOOB, lds = -1;
        if constexpr (XM == 2) {
          const int b = b0 + r;
          lds = (ol * PCH + 4 * r) * PB + 4 * pair;
          if (b < a.B) off = 16 * (b * a.Ci + g * Cig + 2 * pair);
        } else if constexpr (XM == 1) {
          const int b = b0 + r;
          lds = (ol * PCH + r) * PB;
          if (b < a.B) off = 4 * (b * a.Ci + g * Cig);
        } else if constexpr (XM == 3) {
          const int hc = r & 1, q = r >> 1;
          const int img = qpi <= 1 ? q : (int)__umulhi((uint32_t)q, inv_qpi);
          const int rem = q - img * qpi;
          const int yr = W4 <= 1 ? rem : (int)__umulhi((uint32_t)rem, inv_w4);
          const int xq = rem - yr * W4;
          const int b = b0 + img, y = ylo_r + yr * gs_h, x = 4 * xq;
          lds = (ol * PCH + img * PIMG + yr * NXR + x) * PB + hc * 8;
          if (b < a.B) off = 4 * ((b * a.Ci + g * Cig + 4 * hc) * a.HW + y * a.W + x);
        } else if constexpr (XM == 4) {
          const int img = qpi <= 1 ? r : (int)__umulhi((uint32_t)r, inv_qpi);
          const int rem = r - img * qpi;
          const int yr = W4 <= 1 ? rem : (int)__umulhi((uint32_t)rem, inv_w4);
          const int xq = rem - yr * W4;
          const int b = b0 + img, y = ylo_r + yr * gs_h, x = 4 * xq;
          lds = (ol * PCH + img * PIMG + yr * NXR + 2 * xq) * PB;
          if (b < a.B) off = 4 * ((b * a.Ci + g * Cig) * a.HW + y * a.W + x);
        } else {
          const int img = PIMG <= 1 ? r : (int)__umulhi((uint32_t)r, inv_pimg);
          const int rem = r - img * PIMG;
          const int yy = NXR <= 1 ? rem : (int)__umulhi((uint32_t)rem, inv_pw);
          const int xx = rem - yy * NXR;
          const int b = b0 + img, y = y_lo + (yy + kmin_h) * gs_h, x = x_lo + (xx + kmin_w) * gs_w;   // a real pixel by construction
          lds = (ol * PCH + r) * PB;
          if (b < a.B) off = 4 * ((b * a.Ci + g * Cig) * a.HW + y * a.W + x);
        }
        it_off[i] = off;
        it_lds[i] = f < n_items ? lds : -1;
        it_ol[i] = ol;
      }
    }
    if (stamp0) dbg_[214] = __builtin_amdgcn_s_memtime();
    const int HWb = 4 * a.HW;
    const int wave_i0 = wave_u0;  // first item index of this wave (iteration 0)
    const int n_items_w = n_items;

    // One stage = loads (issued one stage AHEAD, into registers that the previous stage has just consumed) -> draws (pure
    // ALU: they run while the loads are in flight) -> sampled weights -> pieces -> LDS -> activations -> pieces -> LDS.
    float xv[PIT][XV];
    auto load_x = [&](int st) {  // every item of stage st
      const int oct0 = st * NO;
#pragma unroll
      for (int i = 0; i < PIT; ++i) {
        if (i == 0 || wave_i0 + kProducers * i < n_items_w) {  // wave-uniform
          const int oc = oct0 + it_ol[i];
          const bool in = it_off[i] != (int)kOOB && oc < G8;  // octets past the end read zeros (their weights are zeros too)
          if constexpr (XM == 2) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const float4 v = ldf4(r_x, guard_off(in, (uint32_t)(it_off[i] + 16 * (8 * oc + c))));
              xv[i][4 * c] = v.x, xv[i][4 * c + 1] = v.y, xv[i][4 * c + 2] = v.z, xv[i][4 * c + 3] = v.w;
            }
          } else if constexpr (XM == 1) {
            const float4 v0 = ldf4(r_x, guard_off(in, (uint32_t)(it_off[i] + 32 * oc))), v1 = ldf4(r_x, guard_off(in, (uint32_t)(it_off[i] + 32 * oc + 16)));
            xv[i][0] = v0.x, xv[i][1] = v0.y, xv[i][2] = v0.z, xv[i][3] = v0.w, xv[i][4] = v1.x, xv[i][5] = v1.y, xv[i][6] = v1.z, xv[i][7] = v1.w;
          } else if constexpr (XM == 3) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float4 v = ldf4(r_x, guard_off(in, (uint32_t)(it_off[i] + (8 * oc + c) * HWb)));
              xv[i][4 * c] = v.x, xv[i][4 * c + 1] = v.y, xv[i][4 * c + 2] = v.z, xv[i][4 * c + 3] = v.w;
            }
          } else if constexpr (XM == 4) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const float4 v = ldf4(r_x, guard_off(in, (uint32_t)(it_off[i] + (8 * oc + c) * HWb)));
              xv[i][4 * c] = v.x, xv[i][4 * c + 1] = v.y, xv[i][4 * c + 2] = v.z, xv[i][4 * c + 3] = v.w;
            }
          } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) xv[i][c] = ldf(r_x, guard_off(in, (uint32_t)(it_off[i] + (8 * oc + c) * HWb)));
          }
        }
      }
    };
    auto sign_bit = [&](uint32_t idx) -> uint32_t {  // bit 31 set: s_in = -1 (hash_sign's stream, element offset in the sample)
      return __float_as_uint(hash_sign(skey_in, idx)) & 0x80000000u;
    };
    auto store_px = [&](char* dst, const float (&v)[8], uint32_t idx0, uint32_t cstride) {  // 8 channels of one pixel -> NP pieces x 16 bytes
      uint32_t ph[8], pm[8], pl[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) split_pieces(v[c], ph[c], pm[c], pl[c]);
      if constexpr (FLIP) {
        uint32_t sb[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) sb[c] = sign_bit(idx0 + (uint32_t)c * cstride);
        *reinterpret_cast<uint4*>(dst + 16 * NP) = make_uint4(pack_hi16(sb[1], sb[0]), pack_hi16(sb[3], sb[2]), pack_hi16(sb[5], sb[4]), pack_hi16(sb[7], sb[6]));
      }
      *reinterpret_cast<uint4*>(dst) = make_uint4(pack_hi16(ph[1], ph[0]), pack_hi16(ph[3], ph[2]), pack_hi16(ph[5], ph[4]), pack_hi16(ph[7], ph[6]));
      *reinterpret_cast<uint4*>(dst + 16) = make_uint4(pack_hi16(pm[1], pm[0]), pack_hi16(pm[3], pm[2]), pack_hi16(pm[5], pm[4]), pack_hi16(pm[7], pm[6]));
      if constexpr (NP == 3)
        *reinterpret_cast<uint4*>(dst + 32) = make_uint4(pack_hi16(pl[1], pl[0]), pack_hi16(pl[3], pl[2]), pack_hi16(pl[5], pl[4]), pack_hi16(pl[7], pl[6]));
    };
    auto store_x = [&](char* Xt, int st) {
#pragma unroll
      for (int i = 0; i < PIT; ++i) {
        if ((i == 0 || wave_i0 + kProducers * i < n_items_w) && it_lds[i] >= 0) {
          char* const dst = Xt + it_lds[i];
          // Flipout: element offset (in the sample's x) of the item's first element -- the index of its sign
          const uint32_t e0 = FLIP ? ((uint32_t)it_off[i] >> 2) + (uint32_t)(8 * (st * NO + it_ol[i])) * (uint32_t)a.HW : 0u;
          if constexpr (XM == 2) {  // 2 channels x the plane's 4 pixels: one dword of each pixel's 16-byte slots
#pragma unroll
            for (int px = 0; px < 4; ++px) {
              uint32_t h0, m0_, l0, h1, m1, l1;
              split_pieces(xv[i][px], h0, m0_, l0);
              split_pieces(xv[i][4 + px], h1, m1, l1);
              *reinterpret_cast<uint32_t*>(dst + px * PB) = pack_hi16(h1, h0);
              *reinterpret_cast<uint32_t*>(dst + px * PB + 16) = pack_hi16(m1, m0_);
              if constexpr (NP == 3) *reinterpret_cast<uint32_t*>(dst + px * PB + 32) = pack_hi16(l1, l0);
              if constexpr (FLIP) *reinterpret_cast<uint32_t*>(dst + px * PB + 16 * NP) = pack_hi16(sign_bit(e0 + 4u + (uint32_t)px), sign_bit(e0 + (uint32_t)px));
            }
          } else if constexpr (XM == 3) {  // 4 pixels x 4 channels: half of each pixel's 16-byte slots
#pragma unroll
            for (int px = 0; px < 4; ++px) {
              uint32_t ph[4], pm[4], pl[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) split_pieces(xv[i][4 * c + px], ph[c], pm[c], pl[c]);
              *reinterpret_cast<uint2*>(dst + px * PB) = make_uint2(pack_hi16(ph[1], ph[0]), pack_hi16(ph[3], ph[2]));
              *reinterpret_cast<uint2*>(dst + px * PB + 16) = make_uint2(pack_hi16(pm[1], pm[0]), pack_hi16(pm[3], pm[2]));
              if constexpr (NP == 3) *reinterpret_cast<uint2*>(dst + px * PB + 32) = make_uint2(pack_hi16(pl[1], pl[0]), pack_hi16(pl[3], pl[2]));
              if constexpr (FLIP) {
                uint32_t sb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) sb[c] = sign_bit(e0 + (uint32_t)(c * a.HW + px));
                *reinterpret_cast<uint2*>(dst + px * PB + 16 * NP) = make_uint2(pack_hi16(sb[1], sb[0]), pack_hi16(sb[3], sb[2]));
              }
            }
          } else if constexpr (XM == 4) {  // input pixels xpar and xpar + 2 of the quad: two patch pixels x 8 channels
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              float v[8];
#pragma unroll
              for (int c = 0; c < 8; ++c) v[c] = xpar ? xv[i][4 * c + 2 * k + 1] : xv[i][4 * c + 2 * k];
              store_px(dst + k * PB, v, e0 + (uint32_t)(2 * k + xpar), (uint32_t)a.HW);
            }
          } else {
            const float v[8] = {xv[i][0], xv[i][1], xv[i][2], xv[i][3], xv[i][4], xv[i][5], xv[i][6], xv[i][7]};
            store_px(dst, v, e0, XM == 1 ? 1u : (uint32_t)a.HW);
          }
        }
      }
    };
    if (NS > 0) load_x(0);
    if (stamp0) dbg_[215] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // cleared buffers
    const bool pstamp = dbg_ && blockIdx.x == 0 && tid == 256;
    for (int st = 0; st <= NS; ++st) {  // NS + 1 barriers, like the consumer arm
      if (pstamp && st < 60) dbg_[128 + 2 * st] = __builtin_amdgcn_s_memtime();
      if (pstamp && st == 3) dbg_[250] = dbg_[251] = __builtin_amdgcn_s_memtime();
      if (st < NS) {
        char* const Wt = wbuf + (st & 1) * W_BYTES;
        char* const Xt = xbuf + (st & 1) * X_BYTES;
        const int oct0 = st * NO;  // first octet of this stage
        // ---- draws (no load feeds them) ----
        float ep[UMAX][4];
#pragma unroll
        for (int i = 0; i < UMAX; ++i)
          if (u_live[i])  // wave-uniform
            philox_normal4(key_w, sample, (e_off[i] + (uint32_t)(8 * oct0)) >> 2, ep[i]);
        if (pstamp && st == 3) dbg_[252] = __builtin_amdgcn_s_memtime();
        // ---- sampled weights -> pieces -> LDS ----
#pragma unroll
        for (int i = 0; i < UMAX; ++i) {
          if (u_live[i]) {  // wave-uniform
            const float m4[4] = {mu[i].x, mu[i].y, mu[i].z, mu[i].w}, s4[4] = {rs[i].x, rs[i].y, rs[i].z, rs[i].w};
            uint32_t wh[4], wm_[4], wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)  // masked units loaded zeros: w = 0 + 0 * eps = 0.  Flipout: the mean alone
              split_pieces(FLIP ? m4[j] : __fadd_rn(m4[j], __fmul_rn(s4[j], ep[i][j])), wh[j], wm_[j], wl[j]);
            if (l_off[i] >= 0) {
              char* const dst = Wt + l_off[i];
              *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(wh[1], wh[0]), pack_hi16(wh[3], wh[2]));
              *reinterpret_cast<uint2*>(dst + W_PIECE) = make_uint2(pack_hi16(wm_[1], wm_[0]), pack_hi16(wm_[3], wm_[2]));
              if constexpr (NP == 3) *reinterpret_cast<uint2*>(dst + 2 * W_PIECE) = make_uint2(pack_hi16(wl[1], wl[0]), pack_hi16(wl[3], wl[2]));
            }
            if constexpr (FLIP) {  // second image: the perturbation sigma * eps
#pragma unroll
              for (int j = 0; j < 4; ++j) split_pieces(__fmul_rn(s4[j], ep[i][j]), wh[j], wm_[j], wl[j]);
              if (l_off[i] >= 0) {
                char* const dst = Wt + W_OP + l_off[i];
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(wh[1], wh[0]), pack_hi16(wh[3], wh[2]));
                *reinterpret_cast<uint2*>(dst + W_PIECE) = make_uint2(pack_hi16(wm_[1], wm_[0]), pack_hi16(wm_[3], wm_[2]));
                *reinterpret_cast<uint2*>(dst + 2 * W_PIECE) = make_uint2(pack_hi16(wl[1], wl[0]), pack_hi16(wl[3], wl[2]));
              }
            }
          }
        }
        if (pstamp && st == 3) dbg_[253] = __builtin_amdgcn_s_memtime();
        // ---- activations: this thread's items of the stage's NO octet planes, split on the way to LDS ----
        store_x(Xt, st);
        if (st + 1 < NS) load_x(st + 1), load_w(st + 1);  // next stage's loads: in flight across the barrier and the draws
      }
      if (pstamp && st < 60) dbg_[128 + 2 * st + 1] = __builtin_amdgcn_s_memtime();
      __syncthreads();
    }
    // bias draw + output-stage constants of this workgroup's channels
    if (ptid < BN) {
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
    if (a.out_vec4 || stage_cols) {  // the consumers pass the output tile through LDS: same barriers and a share of the read-out
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        if (ps > 0) __syncthreads();
        __syncthreads();
        if (a.out_vec4) readout_quads(ps, tid);
        else readout_cols(ps, tid);
      }
    }
  } else {
    // =================================================== CONSUMERS ===========================================================
    // KL sweep of this workgroup's slice of the flat parameter tensors, one float4 group per thread and K-stage, BEHIND the
    // stage's MFMAs: in the layers with many parameters the consumers wait for the producers anyway, in the MFMA-bound layers
    // a slice is a group or two. Same per-thread order of accumulation as one uninterrupted sweep; the partial is published
    // with write-through stores (publish_and_ticket_wt: no L2 write-back, so publishing late costs nothing extra).
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
    auto kl_terms = [&](const float4& m4, const float4& r4, const float4& p4, const float4& q4) {
      const float t0 = kl_term(m4.x, softplus(r4.x), p4.x, q4.x) + kl_term(m4.y, softplus(r4.y), p4.y, q4.y);
      const float t1 = kl_term(m4.z, softplus(r4.z), p4.z, q4.z) + kl_term(m4.w, softplus(r4.w), p4.w, q4.w);
      kl_acc += (double)t0 + (double)t1;
      kl_i += 1024;
    };
    auto kl_group = [&]() {  // -> false when this thread has no whole float4 group left
      if (!(kl_v4 && kl_i + 3 < kl_hi)) return false;
      const float4 m4 = *reinterpret_cast<const float4*>(a.mu_w + kl_i), r4 = *reinterpret_cast<const float4*>(a.rho_w + kl_i);
      const float4 p4 = *reinterpret_cast<const float4*>(a.pmu_w + kl_i), q4 = *reinterpret_cast<const float4*>(a.psig_w + kl_i);
      kl_terms(m4, r4, p4, q4);
      return true;
    };
    auto kl_finish = [&]() {
      while (kl_group()) {}
      for (; kl_i < kl_hi; kl_i += 1024)  // tail quad / unaligned bases
        for (int j = 0; j < 4; ++j)
          if (kl_i + j < kl_hi) kl_acc += (double)kl_term(a.mu_w[kl_i + j], softplus(a.rho_w[kl_i + j]), a.pmu_w[kl_i + j], a.psig_w[kl_i + j]);
      const double wsum = wave_sum(kl_acc);
      // every consumer wave stores its partial write-through and drains the store; the workgroup's ONE ticket is taken by
      // thread 0 behind the next workgroup barrier (kl_ticket below): 256 adds on the counter instead of 1024
      if (lane == 0) __hip_atomic_store(&a.slots[(int)blockIdx.x * 4 + wave], wsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto kl_ticket = [&]() {  // thread 0's wave, after the barrier that follows kl_finish in every consumer wave
      const int nslots = 4 * a.kl_slices;
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
    };
    // Byte address (inside an x buffer) of this lane's operand fragment for every (column group, step): the (octet, tap) entry
    // of (step, lane half) applied to the lane's output pixel -- or the shared zero pixel when that tap falls into the padding
    // for this pixel (or the column / the entry is dead: dead entries meet zero weights, any finite data will do).
    int xaddr[TM][kSplitSteps];
    {
      int pimg[TM], pyb[TM], pxb[TM];
      bool live[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        int b, ho, wo;
        live[j] = col_decode(wm * WTM + j * 32 + li, b, ho, wo);
        pimg[j] = (b - b0) * PIMG, pyb[j] = (ho - r0) * ps_h, pxb[j] = (wo - w0) * ps_w;
      }
#pragma unroll
      for (int q = 0; q < kSplitSteps; ++q) {
        int ol, ai;  // entry of (step q, this lane half)
        if (nA <= 1) ol = 2 * q + lh, ai = 0;
        else ol = (q >= SPO) + (q >= 2 * SPO) + (q >= 3 * SPO) + (q >= 4 * SPO), ai = 2 * (q - ol * SPO) + lh;  // q / SPO, q < 5
        const bool ent = nA > 0 && ol < NO && ai < nA;
        const int4 e = taptab[ent ? ai : 0];
        const int ty = (dymax == dymin) ? 0 : e.y - dymin, tx = (dxmax == dxmin) ? 0 : e.z - dxmin;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int yy = pyb[j] + ty, xx = pxb[j] + tx;  // position on the full patch grid
          const bool ok = ent && live[j] && yy >= kmin_h && yy <= kmax_h && xx >= kmin_w && xx <= kmax_w;
          xaddr[j][q] = ok ? (ol * PCH + pimg[j] + (yy - kmin_h) * NXR + (xx - kmin_w)) * PB : ZOFF;
        }
      }
    }
    int wlq[kSplitSteps];  // this lane's row slot inside the (step, half) plane: swizzled like the producers' writes
#pragma unroll
    for (int q = 0; q < kSplitSteps; ++q) wlq[q] = lh * W_HALF + (li ^ ((2 * q + lh) & 7)) * 16 + wn * TN * 32 * 16;

    f32x16 acc[NOP][TN][TM];
#pragma unroll
    for (int o = 0; o < NOP; ++o)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[o][i][j][r] = 0.f;
    if (prio_mode == 2) __builtin_amdgcn_s_setprio(3);

    const bool cstamp = dbg_ && blockIdx.x == 0 && tid == 0;
    if (cstamp) dbg_[216] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // cleared buffers
    __syncthreads();  // stage 0 staged
    if (cstamp) dbg_[0] = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < NS; ++st) {
      if (cstamp && st < 60) dbg_[2 + 2 * st] = __builtin_amdgcn_s_memtime();
      const char* const Wt = wbuf + (st & 1) * W_BYTES;
      const char* const Xt = xbuf + (st & 1) * X_BYTES;
      int nstep = NSTEP;  // the last stage may hold fewer octets
      if ((st + 1) * NO > G8) {
        const int no_l = G8 - st * NO;
        nstep = nA <= 1 ? (no_l + 1) >> 1 : no_l * SPO;
      }
      // (step, column group) units in a software pipeline: the fragments of unit u+1 (and, at a step's last unit, the next
      // step's weight fragments) are read into the other register set BEFORE the 2*terms MFMAs of unit u, so an LDS round
      // trip never sits between two MFMAs; only the first unit behind the stage barrier waits for its reads.
      {
        constexpr int NXP = NP + (FLIP ? 1 : 0);  // x fragments per unit: the pieces (+ the sign masks)
        bf16x8 wf[2][NOP][TN][NP], xf[2][NXP];
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
          const char* const px = Xt + xaddr[u % TM][u / TM];
#pragma unroll
          for (int p = 0; p < NXP; ++p) xf[u & 1][p] = *reinterpret_cast<const bf16x8*>(px + 16 * p);
        };
        read_w(0);
        read_x(0);
#pragma unroll
        for (int q = 0; q < kSplitSteps; ++q) {
          if (q < nstep) {  // uniform
#pragma unroll
            for (int j = 0; j < TM; ++j) {
              const int u = q * TM + j;
              if (j + 1 < TM) {
                read_x(u + 1);
              } else if (q + 1 < kSplitSteps) {
                if (q + 1 < nstep) {
                  read_x(u + 1);
                  read_w(q + 1);
                }
              }
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int o = 0; o < NOP; ++o) {
                if (o == 1) {  // Flipout: x o s_in -- flip the sign bits of every piece (the split of -v is the negated split of v)
#pragma unroll
                  for (int p = 0; p < NP; ++p) xf[u & 1][p] ^= xf[u & 1][NXP - 1];
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                  // D[pixel][channel]: x is the A operand, W the B operand; terms in decreasing weight
                  acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[u & 1][0], wf[q & 1][o][i][0], acc[o][i][j], 0, 0, 0);
                  acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[u & 1][0], wf[q & 1][o][i][1], acc[o][i][j], 0, 0, 0);
                  acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[u & 1][1], wf[q & 1][o][i][0], acc[o][i][j], 0, 0, 0);
                  if constexpr (NP == 3) {
                    acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[u & 1][0], wf[q & 1][o][i][2], acc[o][i][j], 0, 0, 0);
                    acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[u & 1][1], wf[q & 1][o][i][1], acc[o][i][j], 0, 0, 0);
                    acc[o][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[u & 1][2], wf[q & 1][o][i][0], acc[o][i][j], 0, 0, 0);
                  }
                }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      if (cstamp && st < 60) dbg_[2 + 2 * st + 1] = __builtin_amdgcn_s_memtime();
      if (kl_block) kl_group();   // (fetching the group one stage ahead was tried in round 3: the 16 registers it holds across the MFMA loop spill on
                                  //  the 256-wide tiles -- layer3 +5..19 % -- and layer4's 128-wide launches did not move: dropped)
      __syncthreads();
    }
    if (kl_block) kl_finish();
    if (cstamp) dbg_[1] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // the producers have staged bias / output-stage constants; every consumer wave has published its KL partial
    if (kl_block && wave == 0) kl_ticket();
    if (cstamp) dbg_[120] = __builtin_amdgcn_s_memtime();

    // ---- output stage + store (bt_fused_fast.h: lane = one channel, registers 4q..4q+3 = 4 consecutive positions) ----
    if (a.out_vec4 || stage_cols) {
      constexpr int SROW = BM + 4;
      static_assert((4 * BN + NOP * SROWS * SROW) * 4 <= 2 * (W_BYTES + X_BYTES), "output staging fits the operand buffers");
      float* const stage = smem + 4 * BN;
      float bsv[TN], scv[TN], shv[TN], b1v[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int co_l = (wn * TN + i) * 32 + li;
        bsv[i] = bias0[co_l], scv[i] = osc[co_l], shv[i] = osh[co_l];
        b1v[i] = FLIP ? bias1[co_l] : 0.f;
      }
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        if (ps > 0) __syncthreads();  // the previous pass has been read out
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int cg = wn * TN + i;  // this wave's 32-channel group inside the tile
          if (cg >= ps * (SROWS / 32) && cg < (ps + 1) * (SROWS / 32)) {
            float* const srow = stage + ((cg - ps * (SROWS / 32)) * 32 + li) * SROW + wm * WTM + 4 * lh;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float v[4];
                if constexpr (FLIP) {  // both sets as they are (+ their bias parts): s_out and the constants meet them in the read-out
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
        }
        if (cstamp) dbg_[121] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (cstamp) dbg_[122] = __builtin_amdgcn_s_memtime();
        if (a.out_vec4) readout_quads(ps, tid);
        else readout_cols(ps, tid);
        if (cstamp) dbg_[123] = __builtin_amdgcn_s_memtime();
      }
    } else {
      // Scalar stores: lanes run along the channels (consecutive addresses when Ho*Wo == 1: Linear and 1x1 maps).
      float bsv[TN], scv[TN], shv[TN], b1v[TN];
      bool cok[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int co_l = (wn * TN + i) * 32 + li;
        bsv[i] = bias0[co_l], scv[i] = osc[co_l], shv[i] = osh[co_l];
        b1v[i] = FLIP ? bias1[co_l] : 0.f;
        cok[i] = n0 + co_l < a.Cog;
      }
      const int HoWo = a.Ho * a.Wo;
      auto outv = [&](int i, int j, int r, uint32_t oi) -> float {   // output-stage value of accumulator register r (before residual / ReLU)
        float v = __fadd_rn(acc[0][i][j][r], bsv[i]);
        if constexpr (FLIP) v = __fadd_rn(v, __fmul_rn(__fadd_rn(acc[NOP - 1][i][j][r], b1v[i]), hash_sign(skey_out, oi)));
        return __fadd_rn(__fmul_rn(v, scv[i]), shv[i]);
      };
      // In all three forms below the residual values of a column group are fetched in ONE batch before its first store: written
      // as load -> add -> store per element, the loads cannot be moved above the stores (the compiler has to assume out and
      // residual alias) and every element pays a full memory round trip -- 17 K of layer3's 170 K cycles per workgroup.
      if (a.row_taps && t_Wt == 2 && (a.Wo & 1) == 0) {
        // Row tiles of two-pixel rows: columns 2i, 2i + 1 are the two pixels of one image's row -- 8 contiguous bytes per channel.
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          uint32_t base[8];
          bool live[8];
          float2 rr[8][TN];
#pragma unroll
          for (int r2 = 0; r2 < 8; ++r2) {
            const int r = 2 * r2, row = (r & 3) + 8 * (r >> 2) + 4 * lh;   // even
            int bb, hh, ww;
            live[r2] = col_decode(wm * WTM + j * 32 + row, bb, hh, ww);   // (ww == 0; column + 1 is the same image's second pixel)
            base[r2] = live[r2] ? (uint32_t)((bb * a.Co + g * a.Cog + n0) * HoWo + hh * a.Wo + ww) : 0u;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
              rr[r2][i] = make_float2(0.f, 0.f);
              if (res_s && live[r2] && cok[i]) rr[r2][i] = *reinterpret_cast<const float2*>(res_s + base[r2] + (uint32_t)(((wn * TN + i) * 32 + li) * HoWo));
            }
          }
#pragma unroll
          for (int r2 = 0; r2 < 8; ++r2) {
            const int r = 2 * r2;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
              if (live[r2] && cok[i]) {
                const uint32_t oi = base[r2] + (uint32_t)(((wn * TN + i) * 32 + li) * HoWo);
                float v0 = outv(i, j, r, oi), v1 = outv(i, j, r + 1, oi + 1u);
                if (res_s) v0 = __fadd_rn(v0, rr[r2][i].x), v1 = __fadd_rn(v1, rr[r2][i].y);
                v0 = (relu && v0 < 0.f) ? 0.f : v0, v1 = (relu && v1 < 0.f) ? 0.f : v1;
                *reinterpret_cast<float2*>(out_s + oi) = make_float2(v0, v1);
              }
            }
          }
        }
      } else if (!pix && !a.row_taps && HoWo == 4 && t_R == a.Ho && t_Wt == a.Wo) {
        // Tiles of whole four-pixel images: columns 4i .. 4i + 3 are one image's plane -- 16 contiguous bytes per channel.
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          uint32_t base[4];
          bool live[4];
          float4 rr[4][TN];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 8 * q + 4 * lh;   // a multiple of 4
            int bb, hh, ww;
            live[q] = col_decode(wm * WTM + j * 32 + row, bb, hh, ww);   // (pixel 0 of image bb)
            base[q] = live[q] ? (uint32_t)((bb * a.Co + g * a.Cog + n0) * HoWo) : 0u;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
              rr[q][i] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (res_s && live[q] && cok[i]) rr[q][i] = *reinterpret_cast<const float4*>(res_s + base[q] + (uint32_t)(((wn * TN + i) * 32 + li) * HoWo));
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
              if (live[q] && cok[i]) {
                const uint32_t oi = base[q] + (uint32_t)(((wn * TN + i) * 32 + li) * HoWo);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = outv(i, j, 4 * q + e, oi + (uint32_t)e);
                if (res_s) v[0] = __fadd_rn(v[0], rr[q][i].x), v[1] = __fadd_rn(v[1], rr[q][i].y), v[2] = __fadd_rn(v[2], rr[q][i].z), v[3] = __fadd_rn(v[3], rr[q][i].w);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (relu && v[e] < 0.f) ? 0.f : v[e];
                *reinterpret_cast<float4*>(out_s + oi) = make_float4(v[0], v[1], v[2], v[3]);
              }
            }
          }
        }
      } else
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        uint32_t base[16];
        bool live[16];
        float rr[16][TN];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          int bb, hh, ww;
          live[r] = col_decode(wm * WTM + j * 32 + row, bb, hh, ww);
          base[r] = live[r] ? (uint32_t)((bb * a.Co + g * a.Cog + n0) * HoWo + hh * a.Wo + ww) : 0u;  // channel n0 of this pixel
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            rr[r][i] = 0.f;
            if (res_s && live[r] && cok[i]) rr[r][i] = res_s[base[r] + (uint32_t)(((wn * TN + i) * 32 + li) * HoWo)];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            if (live[r] && cok[i]) {
              const uint32_t oi = base[r] + (uint32_t)(((wn * TN + i) * 32 + li) * HoWo);
              float v = outv(i, j, r, oi);
              if (res_s) v = __fadd_rn(v, rr[r][i]);
              v = (relu && v < 0.f) ? 0.f : v;
              out_s[oi] = v;
            }
          }
        }
      }
    }
  }
  if (dbg_ && blockIdx.x == 0 && tid == 0) dbg_[126] = __builtin_amdgcn_s_memtime();
  if (dbg_ && dbg_[201] && tid == 0) {
    dbg_[256 + 4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    dbg_[256 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime() - dbg_[256 + 4 * blockIdx.x + 2];
  }
}

}  // namespace bt
