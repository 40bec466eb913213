// K1-K5: the fused stochastic forward (+KL) of the four Bayesian layers as ONE kernel template.
//
//   D_s[co][m] = sum_k W_s[co][k] * X_s[k][m]      m = (b, ho, wo),  k = (ci, kh, kw)
//
// Structure (768 threads = 12 waves per workgroup, one workgroup per CU):
//   waves 4-11 PRODUCERS  synthesise the next K-stage into LDS while the consumers multiply the current one:
//       W_s = mu + log1p(exp(rho)) * eps_s  from (mu, rho) [+ injected eps | on-chip Philox4x32-10 + Box-Muller],
//             two roundings exactly like the reference (tmp = sigma*eps; w = mu + tmp); never written to HBM;
//       X_s   the implicit-GEMM (im2col) view of the NCHW input (Linear: the row-major activation tile);
//       Flipout: a second pair of tiles, sigma*eps and x o s_in (signs applied in registers);
//   waves 0-3  CONSUMERS  only read LDS and issue v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: an exact fp32
//             FMA chain -- rtol 1e-4 with K up to 4608 rules out bf16 inputs and gfx950 has no xf32); Flipout keeps two
//             accumulator sets (mean path, perturbation path) over the same staged x tile;
//   the two roles run on the same SIMDs (matrix pipe and VALU are separate pipes), LDS is double-buffered and
//   there is ONE workgroup barrier per K-stage.
// A K-stage is (CC input channels) x (up to 9 ACTIVE taps): taps that can only ever meet zero padding for the
// tile's output pixels are dropped from the schedule -- no loads, no RNG, no MFMA for them (ResNet18/CIFAR layer4
// sees 1x1 images: 1 of 9 taps survives).  With pixel-major m-tiles (one output pixel x 128 images) the same
// pruning applies per pixel (2x2 images: 4 of 9 taps).  Exact: the skipped products are products with zeros.
// KL is swept cooperatively: every workgroup takes a slice of the flat weight tensor (its consumer waves do it
// while the producers fill the first stage), wave-shuffle + LDS reduction, one fp64 slot per workgroup, fixed-order
// finish by the last-arriving workgroup.  MC samples are a grid dimension; (mu, rho) re-reads hit the XCD's L2.
//
// Replaces the ATen chains at reference layers/variational_layers/linear_variational.py:163-181,
// conv_variational.py:366-385, flipout_layers/linear_flipout.py:149-174, conv_flipout.py:376-417.
#pragma once
#include <type_traits>

#include "bt_api_internal.h"

namespace bt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 36;        // K rows of one LDS stage (9 taps x 4 channels)
constexpr int kMaxTaps = 128;  // kh*kw supported by the tap table
#ifndef BT_STAMPS
#define BT_STAMPS 0  // make STAMPS=1: diagnostic build with in-kernel stage stamps
#endif
constexpr bool kStamps = BT_STAMPS != 0;
constexpr int kThreads = 512;   // 4 consumer waves + 4 producer waves: one of each per SIMD
constexpr int kProducers = 256;

struct FwdArgs {
  const float *x, *mu_w, *rho_w, *mu_b, *rho_b, *pmu_w, *psig_w, *pmu_b, *psig_b;
  const float *mu_pk, *sig_pk;  // optional tap-major packed parameters (bt_params.mu_packed / sigma_packed): fast flavour only
  const float *eps_w, *eps_b, *sign_in, *sign_out;
  float* out;
  float* kl_out;
  double* slots;
  unsigned* counter;
  long long x_sample_stride, x_elems, out_elems, w_elems;
  int B, Ci, H, W, Co, KH, KW, SH, SW, PH, PW, DH, DW, G;
  int Ho, Wo, HoWo, M, K, Cig, Cog, S, T, HW;
  int n_tiles, m_tiles, total_blocks;
  int t_NI, t_R, t_Wt, n_bt, n_rt, n_ct;  // fast flavour: tile = t_NI images x t_R rows x t_Wt cols; tile grid per (n-tile, sample)
  int patch_ok;                   // host: tiles are whole images (or pixel-major), so the x operand can be staged as a patch
  int x_cvec;                     // host (fast flavour): x is 16-byte aligned per image -> tiny planes are staged as channel vectors
  int x_rows;                     // host (fast flavour): stage the x patch as 16-byte row chunks written straight to LDS
  int pixel_major, mt_per_pixel;  // m-tile = (one output pixel, BM images) instead of BM consecutive (b, ho, wo)
  int w_vec, x_vec;               // float4 paths allowed (taps == 1, K % 4 == 0, 16-B aligned bases)
  int do_kl, kl_slices;
  uint32_t seed_lo, seed_hi, call, layer_id, sample0;
  const uint32_t* call_base;  // device word added to `call` (fresh draws on graph replay), or null
  const float *ep_scale, *ep_shift, *ep_res;  // fused output stage (bt_epilogue)
  long long ep_res_stride;
  int ep_relu;
  int ep_pool, ep_Hp, ep_Wp;  // fused 3x3 / stride 2 / pad 1 max-pool of the output stage (fast flavour, whole-image tiles)
  int out_vec4;  // spatial output stored as float4 along the pixel index (TRANS orientation; Ho*Wo % 4 == 0, aligned tensors)
  int bn32;                 // general split kernel: 32-channel tiles (launch_split_one)
  unsigned long long* dbg;  // diagnostic stamps (bt_debug_set_stamp_buffer); null in normal operation
  // split flavour: ceil(2^32 / d) of the launch-uniform divisors (0: divide), so the tile decode is a few multiplies
  uint32_t inv_m_tiles, inv_S, inv_n_tiles, inv_n_bt, inv_n_ct, inv_rw, inv_wt, inv_kw;
  int x_flat;  // split flavour, XM 3: the patch is the whole input plane -- fetch it as one row of H*W pixels
  int row_taps;  // split flavour: tiles = t_NI images x ONE output row; the active taps are those of the tile's row (2-row maps)
  // skinny flavour (bt_fused_split_skinny.h): scratch slabs behind the workspace, tickets inside it, slice geometry
  float* sk_scratch;
  unsigned* sk_tickets;
  long long sk_scratch_bytes;
  int sk_nsl, sk_ks, sk_cpt;           // slices per tile, slice width (channels), slices per tap
  int sk_kh0, sk_nh, sk_kw0, sk_nw;    // the rectangle of taps whose input pixel exists for the one output pixel
  int d_tap;     // direct flavour: the ONE tap of the kernel window that meets data (0 for 1x1 kernels; the centre of a padded window over a 1x1 image)
};

// Blocks are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2). Give every XCD a CONTIGUOUS range of
// the logical block order, which is n-tile-major: an XCD then works on few n-tiles for all samples and m-tiles, so
// its (mu, rho) working set stays in its own L2. Bijective for any grid size (cdna_hip_programming.md, T1).
__device__ __forceinline__ int xcd_remap(int orig, int n) {
  const int q = n >> 3, r = n & 7, xcd = orig & 7, i = orig >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// Row-chunk patch (fast flavour, W % 4 == 0): a patch row holds the 16-byte chunks [xa, xa + 4 * n) of an input row, xa = the
// patch's first input column rounded down to a multiple of 4, so every chunk is entirely inside or entirely outside the
// row. When the first chunk and the last are both outside, the last is dropped: those cells alias the next row's first
// chunk, which is zero as well (one spare chunk closes the plane).
__host__ __device__ inline int row_chunks(int x_lo, int x_hi, int W, int* xa_out) {
  const int xa = x_lo & ~3;
  int n = ((x_hi - xa) >> 2) + 1;
  if (n > 1 && xa + 4 * (n - 1) >= W && xa + 4 <= 0) --n;
  *xa_out = xa;
  return n;
}

// Words of one LDS x buffer. Tiles up to 256 columns hold kBK im2col rows; the 512-wide tile (fast flavour only) stages x
// as a patch and gets room for the 2-image 7x7/s2 stem patch of 32x32 inputs (4 x 2 x 37 x 37 words).
// (Flipout keeps two x tiles per stage: its 256-wide tile -- fast flavour only, x as a patch -- gets a 160-column budget:
// room for the 37x37 patch of one 7x7/s2 stem image.)
template <int BM, bool FLIP = false>
constexpr int x_words() {
  return kBK * ((FLIP && BM > 128 ? 160 : BM <= 256 ? BM : 320) + 1);
}

template <int BN, int BM, bool FLIP>
constexpr int fused_lds_bytes() {
  return (2 * (FLIP ? 2 : 1) * (kBK * (BN + 1) + x_words<BM, FLIP>()) + kMaxTaps * 4 + 24 + 8 + 80) * 4;
}

__device__ __forceinline__ double block_sum_all(double v, double* scratch) {  // 12 waves; result in thread 0
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kThreads / 64; ++i) t += scratch[i];
  return t;
}

// INJ: the draws are READ (parity mode) instead of generated; a compile-time switch so that the producers' load phase
// is branch-free straight-line code in both flavours (a runtime branch between two loads is a scheduling barrier).
// LINEAR: row-major [rows][K] operands with K % 4 == 0 and 16-B aligned bases (float4 loads); any other Linear runs as a
// 1x1 convolution over a 1x1 image, which is the same memory layout.
template <int BN, int BM, int CWN, bool FLIP, bool LINEAR, bool TRANS, bool INJ>
__global__ __launch_bounds__(kThreads) void fused_fwd_kernel(const FwdArgs a) {
  unsigned long long* const dbg_ = kStamps ? a.dbg : nullptr;  // stage stamps: diagnostic build only (make STAMPS=1)
  constexpr int CWM = 4 / CWN;
  constexpr int WTN = BN / CWN, WTM = BM / CWM;
  constexpr int TN = WTN / 32, TM = WTM / 32;
  constexpr int WS = BN + 1, XS = BM + 1;  // odd strides: the transposed ds_write_b32 scatter stays (nearly) conflict-free
  constexpr int NW = FLIP ? 2 : 1;
  constexpr int W_WORDS = kBK * WS, X_WORDS = kBK * XS, BUF_WORDS = NW * (W_WORDS + X_WORDS);
  static_assert(TN >= 1 && TM >= 1 && BN % 32 == 0 && BM % 32 == 0 && WTM * CWM == BM && WTN * CWN == BN && BM <= kProducers, "tile shape");
  static_assert(!LINEAR || TRANS, "Linear always stores with lanes along the output features");

  extern __shared__ __attribute__((aligned(16))) float smem[];  // ONE LDS object
  int4* const taptab = reinterpret_cast<int4*>(smem + 2 * BUF_WORDS);
  double* const red = reinterpret_cast<double*>(smem + 2 * BUF_WORDS + kMaxTaps * 4);
  int* const misc = reinterpret_cast<int*>(red + 12);  // [0] active tap count, [1] last-arriver flag, [2..5] active-tap window
  int* const rowtab = misc + 8;                        // [2][40]: LDS offset of every K row of a stage's x tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int ptid = producer ? tid - 256 : tid;  // index inside the role: 512 producer / 256 consumer threads
  const int li = lane & 31, lh = lane >> 5;
  const int cw = wave & 3, wn = cw / CWM, wm = cw % CWM;

  int L = xcd_remap(blockIdx.x, a.total_blocks);
  // (uniform integer division runs on the vector ALU: readfirstlane returns the tile coordinates to scalar registers)
  const int mt = __builtin_amdgcn_readfirstlane(L % a.m_tiles);
  L /= a.m_tiles;
  const int s = __builtin_amdgcn_readfirstlane(L % a.S);
  L /= a.S;
  const int nt = __builtin_amdgcn_readfirstlane(L % a.n_tiles);
  const int g = __builtin_amdgcn_readfirstlane(L / a.n_tiles);
  const int n0 = nt * BN;
  const bool pix = a.pixel_major != 0;
  const int tile_p = pix ? __builtin_amdgcn_readfirstlane(mt / a.mt_per_pixel) : 0;                   // pixel-major: the tile's output pixel
  const int m0 = pix ? (mt - tile_p * a.mt_per_pixel) * BM : mt * BM;  // first image (pixel-major) / first m
  const int m_lim = pix ? a.B : a.M;
  const uint32_t sample = a.sample0 + (uint32_t)s;
  const int K = a.K, T = a.T, Cig = a.Cig;

  RngKey key_w;
  key_w.seed_lo = a.seed_lo;
  key_w.seed_hi = a.seed_hi;
  key_w.call = a.call + (a.call_base ? __builtin_nontemporal_load(a.call_base) : 0u);
  key_w.layer_tensor = layer_tensor_word(a.layer_id, 0);
  uint32_t skey_in = 0, skey_out = 0;
  if (FLIP) {
    RngKey ks = key_w;
    ks.layer_tensor = layer_tensor_word(a.layer_id, 2);
    if (!INJ) skey_in = sign_stream_key(ks, sample);
    ks.layer_tensor = layer_tensor_word(a.layer_id, 3);
    if (!INJ) skey_out = sign_stream_key(ks, sample);
  }

  // ---- active taps of this tile (wave 0: ballot compaction, ascending tap order) ---------------------------------
  if (wave == 0) {
    int base = 0;
    int dy0 = 1 << 20, dy1 = -1, dx0 = 1 << 20, dx1 = -1;  // window of the active taps (input offsets kh*DH, kw*DW)
    for (int t0 = 0; t0 < T; t0 += 64) {
      const int t = t0 + lane;
      bool act = false;
      int4 e = make_int4(0, 0, 0, 0);
      if (t < T) {
        const int kh = t / a.KW, kw = t - kh * a.KW;
        e = make_int4(kh * a.DH * a.W + kw * a.DW, kh * a.DH, kw * a.DW, t);
        if (LINEAR) {
          act = true;
        } else if (pix) {
          const int ho = tile_p / a.Wo, wo = tile_p - ho * a.Wo;
          act = (unsigned)(ho * a.SH - a.PH + e.y) < (unsigned)a.H && (unsigned)(wo * a.SW - a.PW + e.z) < (unsigned)a.W;
        } else {  // some output row/column of the full grid reaches a real input row/column through this tap
          const int lo_h = a.PH - e.y, lo_w = a.PW - e.z;
          const int hc = lo_h > 0 ? (lo_h + a.SH - 1) / a.SH : 0, wc = lo_w > 0 ? (lo_w + a.SW - 1) / a.SW : 0;
          act = hc < a.Ho && hc * a.SH - lo_h < a.H && wc < a.Wo && wc * a.SW - lo_w < a.W;
        }
      }
      const unsigned long long mask = __ballot(act);
      if (act) taptab[base + __popcll(mask & ((1ull << lane) - 1ull))] = e;
      base += __popcll(mask);
      if (act) dy0 = min(dy0, e.y), dy1 = max(dy1, e.y), dx0 = min(dx0, e.z), dx1 = max(dx1, e.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      dy0 = min(dy0, __shfl_xor(dy0, o, 64)), dy1 = max(dy1, __shfl_xor(dy1, o, 64));
      dx0 = min(dx0, __shfl_xor(dx0, o, 64)), dx1 = max(dx1, __shfl_xor(dx1, o, 64));
    }
    if (lane == 0) misc[0] = base, misc[2] = dy0, misc[3] = dy1, misc[4] = dx0, misc[5] = dx1;
  }
  __syncthreads();
  const int nA = __builtin_amdgcn_readfirstlane(misc[0]);

  // ---- stage schedule: (CC channels) x (NA active taps) <= kBK rows -------------------------------------------------
  int NA, CC;
  if (LINEAR || nA == 1) {
    NA = 1, CC = 32;
  } else if (nA == 2) {
    NA = 2, CC = 16;
  } else if (nA <= 4) {
    NA = nA, CC = 8;
  } else {
    NA = nA < 9 ? nA : 9, CC = 4;
  }
  while (CC > 4 && CC / 2 >= Cig) CC >>= 1;  // do not pad tiny channel counts up to a wide chunk
  const int lcc = 31 - __clz(CC);
  const int n_ach = (nA + NA - 1) / NA, n_cch = (Cig + CC - 1) / CC;
  const int NS = nA ? n_ach * n_cch : 0;

  // ---- x operand as an LDS PATCH (conv, whole-image or pixel-major tiles): every input pixel the tile's outputs can
  // touch is staged ONCE per channel (zero halo included) and the consumers address it as base(lane's output pixel) +
  // offset(tap) + channel*plane -- instead of an im2col tile that re-gathers each pixel once per tap. 6-9x fewer load
  // instructions for 3x3 kernels (the texture addresser needs 16 cycles per 64-lane dword load, coalesced or not).
  const int dymin = __builtin_amdgcn_readfirstlane(misc[2]), dymax = __builtin_amdgcn_readfirstlane(misc[3]);
  const int dxmin = __builtin_amdgcn_readfirstlane(misc[4]), dxmax = __builtin_amdgcn_readfirstlane(misc[5]);
  const int t_R = pix ? 1 : a.Ho, t_Wt = pix ? 1 : a.Wo;            // output rows / cols of one image inside the tile
  const int t_NI = pix ? BM : BM / a.HoWo;                            // images per tile
  const int ps_h = (dymax == dymin) ? 1 : a.SH, gs_h = (dymax == dymin) ? a.SH : 1;  // one row offset only: keep just the rows that are read
  const int ps_w = (dxmax == dxmin) ? 1 : a.SW, gs_w = (dxmax == dxmin) ? a.SW : 1;
  const int PHt = (t_R - 1) * ps_h + (dymax - dymin) + 1, PWt = (t_Wt - 1) * ps_w + (dxmax - dxmin) + 1;
  const int PIMG = PHt * PWt, PCH = t_NI * PIMG;                      // plane of one image / of one channel
  const bool use_patch = !LINEAR && a.patch_ok && nA > 0 && CC * PCH <= X_WORDS && PCH < 65536;

  // ---- producer-side constants ----------------------------------------------------------------------------------------
  const float* const xs = a.x + (long long)s * a.x_sample_stride;
  const float* const eps_w_s = INJ ? a.eps_w + (long long)s * a.w_elems : nullptr;
  const float* const sin_s = (FLIP && INJ) ? a.sign_in + (long long)s * a.x_elems : nullptr;
  constexpr int NG = kProducers / BM;  // conv gather: row groups (each thread owns one tile column for 1/NG of the rows)
  const int xm = ptid % BM, xg = ptid / BM;
  int hi0 = 0, wi0 = 0;
  int xoff0 = 0;  // element offsets fit 32 bits (the API rejects tensors of 2^30 elements or more)
  bool mvalid = false;
  if (!LINEAR && producer) {
    const int ml = m0 + xm;
    mvalid = ml < m_lim;
    int b, p;
    if (pix) {
      b = mvalid ? ml : 0, p = tile_p;
    } else {
      const int mm = mvalid ? ml : 0;
      b = mm / a.HoWo, p = mm - b * a.HoWo;
    }
    const int ho = p / a.Wo, wo = p - ho * a.Wo;
    hi0 = ho * a.SH - a.PH;
    wi0 = wo * a.SW - a.PW;
    xoff0 = (b * a.Ci + g * Cig) * a.HW + hi0 * a.W + wi0;
  }
  const int Cig4 = (Cig + 3) & ~3;  // the draw index pads the channel axis to a multiple of 4: a weight unit is one Philox block

  // ---- producer state hoisted out of the stage loop -----------------------------------------------------------------------
  // Weight unit = (row r, channel quad cq, active tap slot ai) -> 4 sampled weights. The (r, cq, ai) of this thread's
  // units depend only on the stage SHAPE (NA taps x CC channels), so they are decoded once; a stage whose tap chunk is
  // short simply masks the slots past its last tap.
  constexpr int UMAX = (BN * 9 + kProducers - 1) / kProducers;  // a stage holds at most 9 quads per row
  const int ncq_ = CC >> 2, lncq_ = lcc - 2;
  const int nunits_full = BN * ncq_ * NA;
  int u_r[UMAX], u_ai[UMAX], u_kc[UMAX];
  uint32_t u_wb[UMAX], u_eb[UMAX];  // row parts of the weight offset and of the draw index
  unsigned u_ok = 0;                // bit i: unit i exists and its row is inside Cog
  if (producer) {
    const uint32_t inv_na = (uint32_t)((0x100000000ull + (unsigned)NA - 1) / (unsigned)NA);
#pragma unroll
    for (int i = 0; i < UMAX; ++i) {
      const int u = ptid + kProducers * i;
      const int uu = u < nunits_full ? u : 0;
      const int tq = NA == 1 ? uu : (int)__umulhi((uint32_t)uu, inv_na);  // uu / NA (exact for uu < 2^16; 2^32/1 does not fit)
      const int ai = uu - tq * NA;
      const int cq = tq & (ncq_ - 1), r = tq >> lncq_;
      const int co_g = n0 + r;
      const uint32_t co = (uint32_t)(g * a.Cog + (co_g < a.Cog ? co_g : 0));
      u_r[i] = r;
      u_ai[i] = ai;
      u_kc[i] = (ai << lcc) + 4 * cq;
      u_wb[i] = co * (uint32_t)K + (uint32_t)(4 * cq * T);
      u_eb[i] = co * (uint32_t)T * (uint32_t)Cig4 + (uint32_t)(4 * cq);
      if (u < nunits_full && co_g < a.Cog) u_ok |= 1u << i;
    }
  }
  const int wave_u0 = __builtin_amdgcn_readfirstlane(ptid & ~63);  // first unit index of this wave (iteration 0)
  // Gather descriptors per tap slot: offset of the tap inside an image plane + "this lane's pixel sees real data".
  int g_off[9];
  unsigned g_ok = 0;
  auto setup_taps = [&](int a0, int na_s) {
    g_ok = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int4 e = taptab[a0 + (t < na_s ? t : 0)];
      g_off[t] = xoff0 + e.x;
      if (mvalid && t < na_s && (unsigned)(hi0 + e.y) < (unsigned)a.H && (unsigned)(wi0 + e.z) < (unsigned)a.W) g_ok |= 1u << t;
    }
  };
  if (!LINEAR && producer && n_ach == 1) setup_taps(0, nA);
  // patch fill: this thread owns plane positions pos = ptid + 256*i (all channels of the stage)
  constexpr int PPOS = (X_WORDS / 4 + kProducers - 1) / kProducers;
  int p_off[PPOS];
  unsigned p_ok = 0;
  if (use_patch && producer) {
    const uint32_t inv_pimg = (uint32_t)((0x100000000ull + (unsigned)PIMG - 1) / (unsigned)PIMG);
    const uint32_t inv_pw = (uint32_t)((0x100000000ull + (unsigned)PWt - 1) / (unsigned)PWt);
    const int y_lo = (pix ? (tile_p / a.Wo) * a.SH : 0) - a.PH + dymin, x_lo = (pix ? (tile_p % a.Wo) * a.SW : 0) - a.PW + dxmin;
    const int b0 = pix ? m0 : m0 / a.HoWo;
#pragma unroll
    for (int i = 0; i < PPOS; ++i) {
      const int pos = ptid + kProducers * i;
      const int pp = pos < PCH ? pos : 0;
      const int img = PIMG == 1 ? pp : (int)__umulhi((uint32_t)pp, inv_pimg);
      const int rem = pp - img * PIMG;
      const int yy = PWt == 1 ? rem : (int)__umulhi((uint32_t)rem, inv_pw);
      const int xx = rem - yy * PWt;
      const int b = b0 + img, y = y_lo + yy * gs_h, x = x_lo + xx * gs_w;
      const bool ok = pos < PCH && b < a.B && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      p_off[i] = ok ? (b * a.Ci + g * Cig) * a.HW + y * a.W + x : 0;
      if (ok) p_ok |= 1u << i;
    }
  }
  // consumers: LDS column base of this lane's output pixel, per 32-wide tile column group
  int colbase[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int ml = wm * WTM + j * 32 + li;
    if (use_patch) {
      const int img = pix ? ml : ml / a.HoWo, p = pix ? 0 : ml - img * a.HoWo;
      const int ho = p / a.Wo, wo = p - ho * a.Wo;
      colbase[j] = img * PIMG + ho * ps_h * PWt + wo * ps_w;
    } else {
      colbase[j] = ml;
    }
  }
  const float* const sig_or_rho = a.rho_w;
  const bool have_sigma = false;  // the general kernel always computes softplus itself

  // One stage of producer work. All global loads of the stage (weights, then activations) are issued before any of
  // them is consumed, unconditionally on clamped offsets and with no select between them (a load under a per-lane
  // condition, or a use of its value before the next load, makes hipcc wait for each one alone); masking happens when
  // the values are written to LDS. The Philox/Box-Muller arithmetic depends on no load and runs while they are in
  // flight. LCC = log2(channels per stage) is a compile-time constant so that the gather is a static (tap, channel) nest.
  auto produce = [&](auto LCCc, int st, float* buf) {
    constexpr int LCC = decltype(LCCc)::value, CCs = 1 << LCC;
    float* const Wt0 = buf;
    float* const Wt1 = buf + W_WORDS;
    float* const Xt0 = buf + NW * W_WORDS;
    float* const Xt1 = Xt0 + X_WORDS;
    const int cch = st / n_ach, ach = st - cch * n_ach;
    const int a0 = ach * NA, c0 = cch * CCs;
    const int na_s = (nA - a0) < NA ? (nA - a0) : NA;
    const bool pst = dbg_ && blockIdx.x == 0 && tid == 256 && st == 3;
    if (pst) dbg_[250] = __builtin_amdgcn_s_memtime();
    if (!LINEAR && n_ach > 1) setup_taps(a0, na_s);
    if (pst) dbg_[239] = __builtin_amdgcn_s_memtime();
    // -------- weights: loads -----------------------------------------------------------------------------------------------
    float mu[UMAX][4], rs[UMAX][4], ep[UMAX][4];
    uint32_t ue0[UMAX];
    unsigned uval[UMAX];  // bit j: element j exists
#pragma unroll
    for (int i = 0; i < UMAX; ++i) {
      uval[i] = 0;
      if (i == 0 || wave_u0 + kProducers * i < nunits_full) {  // wave-uniform: later iterations are mostly empty
        const int tap = LINEAR ? 0 : taptab[a0 + (u_ai[i] < na_s ? u_ai[i] : 0)].w;
        const int ci = c0 + ((u_kc[i]) & (CCs - 1));
        const bool uv = ((u_ok >> i) & 1u) && u_ai[i] < na_s;
        const uint32_t base = u_wb[i] + (uint32_t)(c0 * T + tap);           // natural [co][ci][tap] offset of channel ci
        ue0[i] = u_eb[i] + (uint32_t)(tap * Cig4 + c0);                        // tap-major draw index of channel ci
        unsigned val = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (uv && (ci + j < Cig)) val |= 1u << j;
        uval[i] = val;
        if constexpr (LINEAR) {  // the quad is contiguous and all-in or all-out
          const uint32_t sb = (val & 1u) ? base : 0u;
          const float4 m4 = *reinterpret_cast<const float4*>(a.mu_w + sb);
          const float4 r4 = *reinterpret_cast<const float4*>(sig_or_rho + sb);
          mu[i][0] = m4.x, mu[i][1] = m4.y, mu[i][2] = m4.z, mu[i][3] = m4.w;
          rs[i][0] = r4.x, rs[i][1] = r4.y, rs[i][2] = r4.z, rs[i][3] = r4.w;
          if constexpr (INJ) {
            const float4 e4 = *reinterpret_cast<const float4*>(eps_w_s + sb);
            ep[i][0] = e4.x, ep[i][1] = e4.y, ep[i][2] = e4.z, ep[i][3] = e4.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t sb = ((val >> j) & 1u) ? base + (uint32_t)(j * T) : 0u;
            mu[i][j] = a.mu_w[sb];
            rs[i][j] = sig_or_rho[sb];
            if constexpr (INJ) ep[i][j] = eps_w_s[sb];
          }
        }
      }
    }
    if (pst) dbg_[240] = __builtin_amdgcn_s_memtime();
    // -------- K-row table of this stage's x tile (consumers: address = rowtab[k row] + colbase[lane]) ------------------------
    {
      int* const rt = rowtab + (st & 1) * 40;
      const int KCs = na_s << LCC;
      if (ptid < KCs) {
        int off = ptid * XS;  // im2col / Linear tile: row-major [k][m]
        if (use_patch) {
          const int4 e = taptab[a0 + (ptid >> LCC)];
          off = (ptid & (CCs - 1)) * PCH + (e.y - dymin) * PWt + (e.z - dxmin);
        }
        rt[ptid] = off;
      }
    }
    if (pst) dbg_[241] = __builtin_amdgcn_s_memtime();
    // -------- activations: loads --------------------------------------------------------------------------------------------
    constexpr bool FASTX = !LINEAR && (NG <= 4);                   // static (tap, channel) nest; NG <= 4 <= CC
    constexpr int CPT = FASTX ? CCs / NG : 1;                      // channels per (thread, tap)
    constexpr int TPS = FASTX ? kBK / CCs : 1;                     // tap slots of a stage
    constexpr int RP = kProducers / 8;                             // Linear: tile rows per pass
    constexpr int NXR_G = LINEAR ? ((BM + RP - 1) / RP) * 4 : (FASTX ? TPS * CPT : (kBK + NG - 1) / NG);
    constexpr int NXR_P = LINEAR ? 1 : ((X_WORDS / CCs + kProducers - 1) / kProducers) * CCs;
    constexpr int NXR = NXR_G > NXR_P ? NXR_G : NXR_P;
    static_assert(NXR <= 64, "element mask is 64 bits");
    float xv[NXR], xs_[FLIP ? NXR : 1];
    uint32_t xo[FLIP ? NXR : 1];
    unsigned long long xok = 0;  // bit q: element q is real data (else zero padding / outside the tile)
    if constexpr (LINEAR) {
      const int kq = ptid & 7, mr = ptid >> 3;
#pragma unroll
      for (int p = 0; p < (BM + RP - 1) / RP; ++p) {
        const int rl = mr + p * RP;
        const int m = m0 + rl, k = c0 + 4 * kq;
        const uint32_t off = (uint32_t)m * (uint32_t)K + (uint32_t)k;
        const bool in = rl < BM && m < a.M && k < K;
        if (in) xok |= 0xFull << (4 * p);
        const uint32_t so = in ? off : 0u;
        const float4 x4 = *reinterpret_cast<const float4*>(xs + so);
        xv[4 * p] = x4.x, xv[4 * p + 1] = x4.y, xv[4 * p + 2] = x4.z, xv[4 * p + 3] = x4.w;
        if constexpr (FLIP) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xo[4 * p + j] = off + j;
          if constexpr (INJ) {
            const float4 s4 = *reinterpret_cast<const float4*>(sin_s + so);
            xs_[4 * p] = s4.x, xs_[4 * p + 1] = s4.y, xs_[4 * p + 2] = s4.z, xs_[4 * p + 3] = s4.w;
          }
        }
      }
    } else if (use_patch) {
      constexpr int PC = (X_WORDS / CCs + kProducers - 1) / kProducers;  // plane positions per thread at this channel count
#pragma unroll
      for (int i = 0; i < PC; ++i) {
        if (kProducers * i < PCH) {  // uniform: small planes need few passes
#pragma unroll
          for (int c = 0; c < CCs; ++c) {
            const bool ok = ((p_ok >> i) & 1u) && (c0 + c < Cig);
            if (ok) xok |= 1ull << (i * CCs + c);
            const uint32_t off = ok ? (uint32_t)(p_off[i] + (c0 + c) * a.HW) : 0u;
            xv[i * CCs + c] = xs[off];
            if constexpr (FLIP) {
              xo[i * CCs + c] = off;
              if constexpr (INJ) xs_[i * CCs + c] = sin_s[off];
            }
          }
        }
      }
    } else if constexpr (FASTX) {
      const int ci0 = c0 + xg;
      const int cbase = ci0 * a.HW;
#pragma unroll
      for (int t = 0; t < TPS; ++t) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const bool ok = ((g_ok >> t) & 1u) && (ci0 + c * NG < Cig);
          if (ok) xok |= 1ull << (t * CPT + c);
          const uint32_t off = ok ? (uint32_t)(g_off[t] + cbase + c * NG * a.HW) : 0u;
          xv[t * CPT + c] = xs[off];
          if constexpr (FLIP) {
            xo[t * CPT + c] = off;
            if constexpr (INJ) xs_[t * CPT + c] = sin_s[off];
          }
        }
      }
    } else {
      const int KC = na_s << LCC;
#pragma unroll
      for (int q = 0; q < NXR; ++q) {
        const int kc = xg + q * NG;
        const int4 e = taptab[a0 + (kc < KC ? (kc >> LCC) : 0)];
        const int ci = c0 + (kc & (CCs - 1));
        const bool ok = mvalid && kc < KC && ci < Cig && (unsigned)(hi0 + e.y) < (unsigned)a.H && (unsigned)(wi0 + e.z) < (unsigned)a.W;
        if (ok) xok |= 1ull << q;
        const uint32_t off = ok ? (uint32_t)(xoff0 + ci * a.HW + e.x) : 0u;
        xv[q] = xs[off];
        if constexpr (FLIP) {
          xo[q] = off;
          if constexpr (INJ) xs_[q] = sin_s[off];
        }
      }
    }
    if (pst) dbg_[251] = __builtin_amdgcn_s_memtime();
    // -------- draws (independent of every load above) ----------------------------------------------------------------------
    if constexpr (!INJ) {
#pragma unroll
      for (int i = 0; i < UMAX; ++i)
        if (uval[i]) philox_normal4(key_w, sample, ue0[i] >> 2, ep[i]);
    }
    if (pst) dbg_[252] = __builtin_amdgcn_s_memtime();
    // -------- sampled weights -> LDS (transposed) ---------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < UMAX; ++i) {
      if (i == 0 || wave_u0 + kProducers * i < nunits_full) {
        if (ptid + kProducers * i < nunits_full && u_ai[i] < na_s) {  // every slot of the stage is written (zeros when masked)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ev = (uval[i] >> j) & 1u;
            const float sg = have_sigma ? rs[i][j] : softplus(rs[i][j]);
            const float dl = __fmul_rn(sg, ev ? ep[i][j] : 0.f);
            const float w0 = FLIP ? mu[i][j] : __fadd_rn(mu[i][j], dl);
            Wt0[(u_kc[i] + j) * WS + u_r[i]] = ev ? w0 : 0.f;
            if (FLIP) Wt1[(u_kc[i] + j) * WS + u_r[i]] = ev ? dl : 0.f;
          }
        }
      }
    }
    if (pst) dbg_[253] = __builtin_amdgcn_s_memtime();
    // -------- activations -> LDS -------------------------------------------------------------------------------------------------
    if constexpr (LINEAR) {
      const int kq = ptid & 7, mr = ptid >> 3;
#pragma unroll
      for (int p = 0; p < (BM + RP - 1) / RP; ++p) {
        const int rl = mr + p * RP;
        if (rl < BM) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int idx = (4 * kq + j) * XS + rl;
            const float v = ((xok >> (4 * p + j)) & 1ull) ? xv[4 * p + j] : 0.f;
            Xt0[idx] = v;
            if (FLIP) Xt1[idx] = __fmul_rn(v, INJ ? xs_[4 * p + j] : hash_sign(skey_in, xo[4 * p + j]));
          }
        }
      }
    } else if (use_patch) {
      constexpr int PC = (X_WORDS / CCs + kProducers - 1) / kProducers;
#pragma unroll
      for (int i = 0; i < PC; ++i) {
        const int pos = ptid + kProducers * i;
        if (pos < PCH) {
#pragma unroll
          for (int c = 0; c < CCs; ++c) {
            const float v = ((xok >> (i * CCs + c)) & 1ull) ? xv[i * CCs + c] : 0.f;
            Xt0[c * PCH + pos] = v;
            if (FLIP) Xt1[c * PCH + pos] = __fmul_rn(v, INJ ? xs_[i * CCs + c] : hash_sign(skey_in, xo[i * CCs + c]));
          }
        }
      }
    } else if constexpr (FASTX) {
#pragma unroll
      for (int t = 0; t < TPS; ++t)
#pragma unroll
        for (int c = 0; c < CPT; ++c) {  // rows past the stage's last tap are written too (never read; inside the buffer)
          const int idx = ((t << LCC) + xg + c * NG) * XS + xm;
          const float v = ((xok >> (t * CPT + c)) & 1ull) ? xv[t * CPT + c] : 0.f;
          Xt0[idx] = v;
          if (FLIP) Xt1[idx] = __fmul_rn(v, INJ ? xs_[t * CPT + c] : hash_sign(skey_in, xo[t * CPT + c]));
        }
    } else {
      const int KC = na_s << LCC;
#pragma unroll
      for (int q = 0; q < NXR; ++q) {
        const int kc = xg + q * NG;
        if (kc < KC) {
          const float v = ((xok >> q) & 1ull) ? xv[q] : 0.f;
          Xt0[kc * XS + xm] = v;
          if (FLIP) Xt1[kc * XS + xm] = __fmul_rn(v, INJ ? xs_[q] : hash_sign(skey_in, xo[q]));
        }
      }
    }
  };
  auto produce_stage = [&](int st, float* buf) {
    // (diagnostic stamps 250..253 are written inside produce)
    switch (lcc) {
      case 2: produce(std::integral_constant<int, 2>{}, st, buf); break;
      case 3: produce(std::integral_constant<int, 3>{}, st, buf); break;
      case 4: produce(std::integral_constant<int, 4>{}, st, buf); break;
      default: produce(std::integral_constant<int, 5>{}, st, buf); break;
    }
  };

  auto consume = [&](f32x16 (&acc)[NW][TN][TM], int st, const float* buf) {
    const float* const Wt0 = buf + wn * WTN + li;
    const float* const Wt1 = Wt0 + W_WORDS;
    const float* const Xt0 = buf + NW * W_WORDS;
    const float* const Xt1 = Xt0 + X_WORDS;
    const int* const rt = rowtab + (st & 1) * 40 + lh;
    const int cch = st / n_ach, ach = st - cch * n_ach;
    const int na_s = (nA - ach * NA) < NA ? (nA - ach * NA) : NA;
    const int KC = na_s << lcc;  // a multiple of 4 (CC >= 4)
    // fragments of step kk + 2 are read from LDS while the MFMAs of step kk execute
    float af[2][NW][TN], bf[2][NW][TM];
    auto load_frags = [&](auto slotc, int kk) {
      constexpr int slot = decltype(slotc)::value;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        af[slot][0][i] = Wt0[(kk + lh) * WS + i * 32];
        if (FLIP) af[slot][NW - 1][i] = Wt1[(kk + lh) * WS + i * 32];
      }
      const int ro = rt[kk];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        bf[slot][0][j] = Xt0[ro + colbase[j]];
        if (FLIP) bf[slot][NW - 1][j] = Xt1[ro + colbase[j]];
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
    constexpr std::integral_constant<int, 0> s0{};
    constexpr std::integral_constant<int, 1> s1{};
    load_frags(s0, 0);
    for (int kk = 0; kk < KC; kk += 4) {  // KC is a multiple of 4
      load_frags(s1, kk + 2);
      mfmas(s0);
      if (kk + 4 < KC) load_frags(s0, kk + 4);
      mfmas(s1);
    }
  };

  float* const buf0 = smem;
  float* const buf1 = smem + BUF_WORDS;
  float* const bias0 = smem;       // reparam: mu_b + sigma_b*eps_b ; flipout: mu_b   (aliases stage memory after the last barrier)
  float* const bias1 = smem + BN;  // flipout: sigma_b*eps_b
  float* const osc = smem + 2 * BN;  // output stage: per-channel scale (1 when absent)
  float* const osh = smem + 3 * BN;  //               per-channel shift (0 when absent)
  const bool kl_block = a.do_kl && (int)blockIdx.x < a.kl_slices;

  // The two roles are separate control-flow arms with the same number of workgroup barriers, so the accumulator
  // registers live only in the consumer arm and the producer arm gets the whole register budget for loads in flight.
  if (producer) {
    const bool stamp = dbg_ && blockIdx.x == 0 && tid == 256;
    for (int st = 0; st <= NS; ++st) {  // NS + 1 barriers, like the consumer arm
      if (stamp && st < 60) dbg_[128 + 2 * st] = __builtin_amdgcn_s_memtime();
      if (st < NS) produce_stage(st, (st & 1) ? buf1 : buf0);
      if (stamp && st < 60) dbg_[128 + 2 * st + 1] = __builtin_amdgcn_s_memtime();
      __syncthreads();
    }
    if (stamp) dbg_[127] = __builtin_amdgcn_s_memtime();
    // bias draw for this workgroup's output channels
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
  } else {
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
    __syncthreads();

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
    for (int st = 0; st < NS; ++st) {  // one barrier per stage
      if (stamp && st < 60) dbg_[2 + 2 * st] = __builtin_amdgcn_s_memtime();
      consume(acc, st, (st & 1) ? buf1 : buf0);
      if (stamp && st < 60) dbg_[2 + 2 * st + 1] = __builtin_amdgcn_s_memtime();
      __syncthreads();
    }
    if (stamp) dbg_[1] = __builtin_amdgcn_s_memtime();
    __syncthreads();  // the producers have staged the bias

    // output stage + store
    float* const out_s = a.out + (long long)s * a.out_elems;
    const float* const sout_s = (FLIP && INJ) ? a.sign_out + (long long)s * a.out_elems : nullptr;
    const float* const res_s = a.ep_res ? a.ep_res + (long long)s * a.ep_res_stride : nullptr;
    const bool relu = a.ep_relu != 0;
    if (TRANS && a.out_vec4) {
      // spatial NCHW output through the D[m][co] orientation: a lane owns ONE output channel (bias / scale / shift are lane
      // constants) and registers 4q..4q+3 are 4 consecutive output positions -> one 16-byte store (and residual load) per 4
      // values. The host guarantees Ho*Wo % 4 == 0 and 16-byte aligned tensors, so a quad never straddles an image.
#pragma unroll
      for (int j = 0; j < TM; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ml = m0 + wm * WTM + j * 32 + 8 * q + 4 * lh;
          const int bq = ml / a.HoWo, pq = ml - bq * a.HoWo;
          const bool mok = ml < m_lim;
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            const int co_l = wn * WTN + i * 32 + li;
            const bool ok = mok && n0 + co_l < a.Cog;
            const uint32_t oidx = ok ? (uint32_t)((bq * a.Co + g * a.Cog + n0 + co_l) * a.HoWo + pq) : 0u;
            const float b0 = bias0[co_l], sc = osc[co_l], sh = osh[co_l];
            float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (res_s) r4 = *reinterpret_cast<const float4*>(res_s + oidx);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = __fadd_rn(acc[0][i][j][4 * q + e], b0);
              if constexpr (FLIP) {
                const float so = INJ ? sout_s[oidx + e] : hash_sign(skey_out, oidx + e);
                v[e] = __fadd_rn(v[e], __fmul_rn(__fadd_rn(acc[NW - 1][i][j][4 * q + e], bias1[co_l]), so));
              }
              v[e] = __fadd_rn(__fmul_rn(v[e], sc), sh);
            }
            v[0] = __fadd_rn(v[0], r4.x), v[1] = __fadd_rn(v[1], r4.y), v[2] = __fadd_rn(v[2], r4.z), v[3] = __fadd_rn(v[3], r4.w);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (relu && v[e] < 0.f) ? 0.f : v[e];
            if (ok) *reinterpret_cast<float4*>(out_s + oidx) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    } else {
  #pragma unroll
      for (int j = 0; j < TM; ++j) {
        int b_col = 0, p_col = tile_p;
        if (!TRANS) {  // lanes run along m: decode this lane's column once
          const int ml = m0 + wm * WTM + j * 32 + li;
          if (pix) {
            b_col = ml;
          } else {
            b_col = ml / a.HoWo;
            p_col = ml - b_col * a.HoWo;
          }
        }
  #pragma unroll
        for (int i = 0; i < TN; ++i) {
          // All reads of the tile (residual, injected signs) are issued before the first dependent instruction:
          // unconditional loads on clamped indices, selected afterwards (same reason as in produce()).
          uint32_t oi[16];
          bool okv[16];
          int col[16];
  #pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            int co_l, ml;
            if (TRANS) {  // D[m][co]: lanes along co (HoWo == 1, or pixel-major tiles)
              co_l = wn * WTN + i * 32 + li;
              ml = m0 + wm * WTM + j * 32 + row;
              oi[r] = (uint32_t)((ml * a.Co + g * a.Cog + n0 + co_l) * a.HoWo + tile_p);
            } else {      // D[co][m]: lanes along the spatial index (NCHW-contiguous)
              co_l = wn * WTN + i * 32 + row;
              ml = m0 + wm * WTM + j * 32 + li;
              oi[r] = (uint32_t)((b_col * a.Co + g * a.Cog + n0 + co_l) * a.HoWo + p_col);
            }
            okv[r] = n0 + co_l < a.Cog && ml < m_lim;
            col[r] = co_l;
            if (!okv[r]) oi[r] = 0u;
          }
          float rs[16], so[FLIP ? 16 : 1];
          if (res_s) {
  #pragma unroll
            for (int r = 0; r < 16; ++r) rs[r] = res_s[oi[r]];
          } else {
  #pragma unroll
            for (int r = 0; r < 16; ++r) rs[r] = 0.f;
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
            v = __fadd_rn(v, rs[r]);
            v = (relu && v < 0.f) ? 0.f : v;  // a select, so NaN propagates like torch's relu
            if (okv[r]) out_s[oi[r]] = v;
          }
        }
      }
    }
  }

  if (dbg_ && blockIdx.x == 0 && tid == 0) dbg_[126] = __builtin_amdgcn_s_memtime();
}

}  // namespace bt
