// Host-side helpers shared by the split flavour's translation units (bt_fused_split.hip, bt_fused_split_flip.hip).
#pragma once
#include <stdlib.h>

#include "bt_fused_split.h"

namespace bt {

int contraction_mode();  // bt_fused_split.hip: 0 automatic, 1 fp32 MFMA only, 2 bf16x2 (opt-in)

// ceil(2^32 / d): __umulhi(n, .) == n / d for every dividend n with n * d < 2^32. 0 when that cannot be promised (or d == 1):
// the kernel then divides.
static inline uint32_t inv_u32(long long d, long long nmax) {
  if (d <= 1 || nmax < 0 || (unsigned long long)nmax * (unsigned long long)d >= (1ull << 32)) return 0u;
  return (uint32_t)(((1ull << 32) + (unsigned long long)d - 1ull) / (unsigned long long)d);
}
// Reciprocals of the launch-uniform divisors of the tile decode (every workgroup used to spend ~2,000 cycles dividing).
static inline void split_fill_inverses(FwdArgs& a) {
  static const bool off = getenv("BT_NO_HOST_INV") != nullptr;   // test hook: every kernel-side division takes its fallback path
  if (off) {
    a.inv_m_tiles = a.inv_S = a.inv_n_tiles = a.inv_n_bt = a.inv_n_ct = a.inv_rw = a.inv_wt = a.inv_kw = 0u;
    return;
  }
  const long long tb = a.total_blocks;
  a.inv_m_tiles = inv_u32(a.m_tiles, tb);
  a.inv_S = inv_u32(a.S, tb);
  a.inv_n_tiles = inv_u32(a.n_tiles, tb);
  a.inv_n_bt = inv_u32(a.n_bt, a.m_tiles);
  a.inv_n_ct = inv_u32(a.n_ct, a.m_tiles);
  a.inv_rw = inv_u32((long long)a.t_R * a.t_Wt, 1024);
  a.inv_wt = inv_u32(a.t_Wt, 1024);
  a.inv_kw = inv_u32(a.KW, 64);
}

// Extent of the window of taps that can meet data along one axis (the kernel's own rule: bt_fused_split.h), for the whole
// output axis or -- pixel-major tiles prune per pixel -- the widest window of any single output position.
static void tap_window(int K, int D, int S, int P, int In, int Out, bool per_pixel, int* n_act, int* extent, int* n_min = nullptr) {
  int best_n = 0, best_ext = 0, least_n = 1 << 30;
  if (per_pixel) {
    for (int o = 0; o < Out; ++o) {
      int lo = 1 << 30, hi = -1, n = 0;
      for (int k = 0; k < K; ++k)
        if ((unsigned)(o * S - P + k * D) < (unsigned)In) lo = k * D < lo ? k * D : lo, hi = k * D > hi ? k * D : hi, ++n;
      if (n > best_n) best_n = n;
      if (n < least_n) least_n = n;
      if (hi - lo > best_ext) best_ext = hi - lo;
    }
    if (n_min) *n_min = least_n;
  } else {
    int lo = 1 << 30, hi = -1;
    for (int k = 0; k < K; ++k) {
      const int l = P - k * D, c = l > 0 ? (l + S - 1) / S : 0;
      if (c < Out && c * S - l < In) lo = k * D < lo ? k * D : lo, hi = k * D > hi ? k * D : hi, ++best_n;
    }
    best_ext = hi >= lo ? hi - lo : 0;
  }
  *n_act = best_n, *extent = best_ext;
}

// One axis of a tile that spans the whole output axis: how many input positions does the patch keep (the kernel's own rule,
// bt_fused_split.h: the window of the active taps on a grid of spacing gs = 1, or the stride when a single tap is active)?
static int split_axis_kept(int K, int D, int S, int P, int In, int Out, int* gs_out) {
  int lo = 1 << 30, hi = -1;
  for (int k = 0; k < K; ++k) {
    const int l = P - k * D, c = l > 0 ? (l + S - 1) / S : 0;
    if (c < Out && c * S - l < In) lo = k * D < lo ? k * D : lo, hi = k * D > hi ? k * D : hi;
  }
  *gs_out = 1;
  if (hi < 0) return 0;
  const int ext = hi - lo, gs = ext ? 1 : S, ps = ext ? S : 1;
  const int x_lo = -P + lo, Pt = (Out - 1) * ps + ext + 1;
  const int kmin = x_lo < 0 ? (-x_lo + gs - 1) / gs : 0;
  int kmax = In - 1 - x_lo >= 0 ? (In - 1 - x_lo) / gs : -1;
  if (kmax > Pt - 1) kmax = Pt - 1;
  *gs_out = gs;
  return kmax >= kmin ? kmax - kmin + 1 : 0;
}
// Tiles of whole output rows (t_Wt == Wo): which columns of its input rows does the patch keep? 3: every column (the x fetch can
// move 16-byte row pieces, XM 3), 4: every second column (XM 4), 0: neither / W % 4 != 0.
static int split_row_mode(const FwdArgs& a) {
  int gs;
  const int kept = split_axis_kept(a.KW, a.DW, a.SW, a.PW, a.W, a.Wo, &gs);
  if (gs > 2 || (a.W & 3)) return 0;
  if (gs == 1) return kept == a.W ? 3 : 0;
  return kept == a.W / 2 ? 4 : 0;
}
static bool split_rows_cover(const FwdArgs& a) { return split_row_mode(a) == 3; }
// Tiles of whole images whose patch is the whole input plane (every row, every column): a plane is then one contiguous run of
// H*W floats in memory AND in the patch, so with H*W % 4 == 0 the 16-byte fetch of XM 3 works on the flattened plane whatever
// W is (ResNet50's 14x14 maps).
static bool split_plane_flat(const FwdArgs& a) {
  if ((a.HW & 3) || a.t_R != a.Ho || a.t_Wt != a.Wo) return false;
  int gh, gw;
  const int kh = split_axis_kept(a.KH, a.DH, a.SH, a.PH, a.H, a.Ho, &gh), kw = split_axis_kept(a.KW, a.DW, a.SW, a.PW, a.W, a.Wo, &gw);
  return gh == 1 && gw == 1 && kh == a.H && kw == a.W;
}

// Tile geometry as bt_fused_dispatch.h's fast_geometry, with the split flavour's capacity: the patch of ONE octet plane has to
// fit XPO pixels. Fills the tile fields and returns the tile's live columns (0: does not fit).
template <int BM, bool FLIP = false>
static int split_geometry(FwdArgs& a) {
  int nh, nw, dys, dxs, nh_min = 0, nw_min = 0;
  tap_window(a.KH, a.DH, a.SH, a.PH, a.H, a.Ho, a.pixel_major != 0 || a.row_taps != 0, &nh, &dys, &nh_min);   // (row tiles: the window of ONE output row)
  tap_window(a.KW, a.DW, a.SW, a.PW, a.W, a.Wo, a.pixel_major != 0, &nw, &dxs, &nw_min);
  // One active tap: the canonical K order pairs consecutive octets in one MFMA step, so a stage has to hold TWO octet planes
  // whatever the tile (otherwise the pairing, and with it the rounding, would depend on the tile choice). Pixel-major tiles
  // prune per pixel: the rule applies when some pixel is left with a single tap.
  const bool one_tap = a.pixel_major ? nh_min * nw_min <= 1 : nh * nw <= 1;
  const long long XPO = one_tap ? (split_xpo<BM, FLIP>() - 1) / 2 : split_xpo<BM, FLIP>() - 1;   // (one slot is the shared zero pixel)
  auto fits = [&](int NI, int R, int Wt) {
    // the patch stores only pixels that exist: at most the window's rows / columns, at most the image's (on the patch grid)
    long long PHt = (long long)(R - 1) * (dys ? a.SH : 1) + dys + 1, PWt = (long long)(Wt - 1) * (dxs ? a.SW : 1) + dxs + 1;
    const long long rows_max = dys ? a.H : (a.H - 1) / a.SH + 1, cols_max = dxs ? a.W : (a.W - 1) / a.SW + 1;
    if (PHt > rows_max) PHt = rows_max;
    if (PWt > cols_max) PWt = cols_max;
    return NI * PHt * PWt <= XPO;
  };
  int NI, R, Wt;
  if (a.row_taps) {   // images x one output row (bt_fused_split.hip: 2-row maps)
    NI = BM / a.Wo, R = 1, Wt = a.Wo;
    if (NI > a.B) NI = a.B;
    if (NI < 1 || !fits(NI, R, Wt)) return 0;
  } else if (a.HoWo == 1 || a.pixel_major) {
    NI = BM, R = 1, Wt = 1;
    if (NI > a.B) NI = a.B;
    if (!fits(NI, R, Wt)) return 0;
  } else if (a.HoWo <= BM) {
    NI = BM / a.HoWo, R = a.Ho, Wt = a.Wo;  // whole images
    if (NI > a.B) NI = a.B;
    while (NI > 1 && !fits(NI, R, Wt)) --NI;
    if (!fits(NI, R, Wt)) return 0;
  } else if (a.Wo <= BM) {
    NI = 1, R = BM / a.Wo, Wt = a.Wo;  // a band of rows of one image
    while (R > 1 && !fits(NI, R, Wt)) --R;
    if (!fits(NI, R, Wt)) return 0;
  } else {
    NI = 1, R = 1, Wt = BM;  // a segment of one row
    if (!fits(NI, R, Wt)) return 0;
  }
  a.t_NI = NI, a.t_R = R, a.t_Wt = Wt;
  a.n_bt = (a.B + NI - 1) / NI;
  a.n_rt = (a.pixel_major || a.row_taps) ? a.Ho : (a.HoWo > 1 ? (a.Ho + R - 1) / R : 1);
  a.n_ct = a.pixel_major ? a.Wo : (a.HoWo > 1 ? (a.Wo + Wt - 1) / Wt : 1);
  a.m_tiles = a.n_bt * a.n_rt * a.n_ct;
  return NI * R * Wt;
}

}  // namespace bt
