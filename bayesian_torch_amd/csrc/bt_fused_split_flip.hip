// Host side of the split-precision Flipout forward (bt_fused_split.h, FLIP = true): eligibility, tile geometry, launch.
#include "bt_fused_split_quad.h"
#include "bt_fused_split_host.h"

namespace bt {

template <int BM, int NPW, int XM>
static int launch_split_flip_cfg(FwdArgs& a, hipStream_t stream) {
  constexpr int BN = 64, NP = 3;
  constexpr int lds = split_lds_bytes<BN, BM, NP, true>();
  static_assert(lds <= 160 * 1024, "LDS budget of one CU");
  auto kern = fused_split_kernel<BN, BM, NP, NPW, XM, true>;
  static bool flags[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "fused forward (split, flipout): hipGetDevice failed");
  if (!flags[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return set_error(BT_ERR_HIP_BASE, "fused forward (split, flipout): cannot raise the dynamic LDS limit");
    flags[dev] = true;
  }
  char nm[160];
  snprintf(nm, sizeof(nm), "fused_split_kernel<%d,%d,bf16x%d,2x6 terms,flip,npw=%d,xm=%d>", BN, BM, NP, NPW, XM);
  note_kernel(nm);
  split_fill_inverses(a);
  hipLaunchKernelGGL(kern, dim3((unsigned)a.total_blocks), dim3(256 + 64 * NPW), lds, stream, a);
  return check_launch("fused forward (split, flipout)");
}

// Flipout stems (<= 3 input channels per group): bt_fused_split_quad.h with FLIP = true, 64 x 256 tiles of whole images or of
// bands of whole rows; the patch has to fit the 1600 pixels the two weight images leave.
template <bool POOL>
static int launch_quad_flip_cfg(FwdArgs& a, hipStream_t stream) {
  constexpr int lds = quad_lds_bytes<true>();
  static_assert(lds <= 160 * 1024, "LDS budget of one CU");
  auto kern = fused_split_quad_kernel<3, POOL, true>;
  static bool flags[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "fused forward (split, flipout): hipGetDevice failed");
  if (!flags[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return set_error(BT_ERR_HIP_BASE, "fused forward (split, flipout): cannot raise the dynamic LDS limit");
    flags[dev] = true;
  }
  note_kernel(POOL ? "fused_split_quad_kernel<64,256,bf16x3,2x6 terms,flip,pool=1>" : "fused_split_quad_kernel<64,256,bf16x3,2x6 terms,flip,pool=0>");
  split_fill_inverses(a);
  hipLaunchKernelGGL(kern, dim3((unsigned)a.total_blocks), dim3(512), lds, stream, a);
  return check_launch("fused forward (split, flipout, quad)");
}

static int launch_quad_flip(FwdArgs& a, hipStream_t stream) {
  if (a.Cig > 3 || a.pixel_major || a.T > 64 || !a.out_vec4 || a.HoWo < 2 || a.Wo > 256) return 1;
  int nh, nw, dys, dxs;
  tap_window(a.KH, a.DH, a.SH, a.PH, a.H, a.Ho, false, &nh, &dys);
  tap_window(a.KW, a.DW, a.SW, a.PW, a.W, a.Wo, false, &nw, &dxs);
  constexpr long long XCAP = kQuadXBytesFlip / 24;
  const long long PWt = (long long)(a.Wo - 1) * (dxs ? a.SW : 1) + dxs + 1;
  auto rows_px = [&](int R) { return ((long long)(R - 1) * (dys ? a.SH : 1) + dys + 1) * PWt; };
  int NI, R;
  long long tiles_per_sample;
  if (a.HoWo <= 256) {   // whole images
    NI = 256 / a.HoWo, R = a.Ho;
    if (NI > a.B) NI = a.B;
    while (NI > 1 && NI * rows_px(R) > XCAP) --NI;
    if (NI * rows_px(R) > XCAP) return 1;
    tiles_per_sample = (a.B + NI - 1) / NI;
  } else {               // a band of whole rows of one image
    if (a.ep_pool) return 1;
    NI = 1, R = 256 / a.Wo;
    while (R > 1 && rows_px(R) > XCAP) --R;
    if (rows_px(R) > XCAP) return 1;
    tiles_per_sample = (long long)a.B * ((a.Ho + R - 1) / R);
  }
  if ((double)a.M / ((double)tiles_per_sample * 256) < 0.75) return 1;   // the tile must be filled
  if (a.ep_pool) {
    const int Wp = a.ep_Wp;
    if ((Wp & (Wp - 1)) != 0 || Wp < 4 || Wp > 16 || a.ep_res) return 1;
  }
  FwdArgs b = a;
  b.n_tiles = (b.Cog + 63) / 64;
  b.t_NI = NI, b.t_R = R, b.t_Wt = b.Wo, b.n_bt = (b.B + NI - 1) / NI, b.n_rt = (b.Ho + R - 1) / R, b.n_ct = 1, b.m_tiles = b.n_bt * b.n_rt;
  const long long total = (long long)b.G * b.n_tiles * b.S * b.m_tiles;
  if (total <= 0 || total > 0x7FFFFFFFll) return 1;
  b.total_blocks = (int)total;
  b.kl_slices = total < 256 ? (int)total : 256;
  const int rc = b.ep_pool ? launch_quad_flip_cfg<true>(b, stream) : launch_quad_flip_cfg<false>(b, stream);
  if (rc == BT_OK) a = b;
  return rc;
}

// Returns BT_OK when the launch was taken, 1 when this flavour does not apply (the caller runs the fp32 kernels), < 0 on error.
// Tiles: 64 channels x 256 output positions of whole images / row bands (the two accumulator sets of Flipout fill the
// consumers' registers at 32 x 128 per wave), or x 128 (the small feature maps: pixel-major tiles prune the padding taps per
// pixel, 1x1 maps); the patch of one octet plane has to fit 301 pixels (two planes when a single tap is active).
int launch_split_flip(FwdArgs& a, hipStream_t stream) {
  if (contraction_mode() != 0) return 1;   // f32: the fp32 kernels; bf16x2: Reparameterization only
  if (!a.mu_pk || (((uintptr_t)a.mu_pk | (uintptr_t)a.sig_pk) & 15u) || a.w_elems >= (1ll << 29) || a.x_elems >= (1ll << 29)) return 1;
  if (a.Cig <= 4) return launch_quad_flip(a, stream);   // the stems
  if ((a.Cig & 7) || a.T > 9 || a.ep_pool) return 1;
  const int Mdom = a.pixel_major ? a.B : a.M;
  if (Mdom < 112) return 1;
  a.n_tiles = (a.Cog + 63) / 64;
  FwdArgs b256 = a, b128 = a;
  const int live256 = (Mdom >= 256 && !a.pixel_major) ? split_geometry<256, true>(b256) : 0;
  const int live128 = split_geometry<128, true>(b128);
  const long long per = (long long)a.G * a.n_tiles * a.S;
  auto cost = [&](int live, int BM, const FwdArgs& b) -> double {   // as launch_split_one (bt_fused_split.hip)
    if (!live) return 1e30;
    const double eff = (double)a.M / ((double)b.m_tiles * BM);
    if (eff < 0.75) return 1e30;
    const double rounds = (double)((per * b.m_tiles + 255) / 256);
    return rounds * ((BM > 128 ? BM : 128) + 48);
  };
  const double c256 = cost(live256, 256, b256), c128 = cost(live128, 128, b128);
  int bm = 0;
  if (c256 < 1e30 && c256 <= c128) bm = 256;
  else if (c128 < 1e30) bm = 128;
  if (!bm) return 1;
  FwdArgs b = bm == 256 ? b256 : b128;
  const long long total = per * b.m_tiles;
  if (total <= 0 || total > 0x7FFFFFFFll) return 1;
  b.total_blocks = (int)total;
  b.kl_slices = total < 256 ? (int)total : 256;
  const bool xal = (((uintptr_t)b.x) & 15u) == 0 && (b.x_sample_stride & 3) == 0;
  int rc;
  if (bm == 256) {
    const bool rows = xal && b.HW > 1 && b.SH == 1 && b.SW == 1 && (b.W & 3) == 0 && b.t_Wt == b.Wo && split_rows_cover(b);
    rc = rows ? launch_split_flip_cfg<256, 4, 3>(b, stream) : launch_split_flip_cfg<256, 4, 0>(b, stream);
  } else {
    if (xal && b.HW == 1) rc = launch_split_flip_cfg<128, 8, 1>(b, stream);
    else if (xal && b.pixel_major && b.H == 2 && b.W == 2 && b.KH == 3 && b.KW == 3 && b.PH == 1 && b.PW == 1 && b.SH == 1 && b.SW == 1 && b.DH == 1 && b.DW == 1)
      rc = launch_split_flip_cfg<128, 8, 2>(b, stream);
    else rc = launch_split_flip_cfg<128, 8, 0>(b, stream);
  }
  if (rc == BT_OK) a = b;
  return rc;
}

}  // namespace bt
