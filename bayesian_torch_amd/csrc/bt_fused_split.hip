// Host side of the split-precision flavour (bt_fused_split.h): eligibility, tile geometry, launch.
#include <atomic>
#include <stdlib.h>
#include <string.h>

#include "bt_fused_split.h"

namespace bt {

// 0: automatic (6-term exact split where the launch is eligible), 1: fp32 MFMA only, 2: 3-term split (opt-in, ~1e-5 relative)
static std::atomic<int> g_contraction{-1};
static int contraction_mode() {
  int m = g_contraction.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("BT_CONTRACTION");
    m = !e ? 0 : (!strcmp(e, "f32") ? 1 : !strcmp(e, "bf16x2") ? 2 : 0);
    g_contraction.store(m, std::memory_order_relaxed);
  }
  return m;
}

// Tile geometry as bt_fused_dispatch.h's fast_geometry, with the split flavour's capacity: the patch of ONE octet plane
// (worst case: every tap active) has to fit XPO pixels. Returns the number of live columns of a tile (0: does not fit).
template <int BM>
static int split_geometry(FwdArgs& a) {
  constexpr long long XPO = split_xpo<BM>();
  const int dys = (a.KH - 1) * a.DH, dxs = (a.KW - 1) * a.DW;
  auto fits = [&](int NI, int R, int Wt) {
    const long long PHt = (long long)(R - 1) * (dys ? a.SH : 1) + dys + 1, PWt = (long long)(Wt - 1) * (dxs ? a.SW : 1) + dxs + 1;
    return NI * PHt * PWt <= XPO;
  };
  int NI, R, Wt;
  if (a.HoWo == 1 || a.pixel_major) {
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
  const bool grid = a.pixel_major || a.HoWo > 1;
  a.t_NI = NI, a.t_R = R, a.t_Wt = Wt;
  a.n_bt = (a.B + NI - 1) / NI;
  a.n_rt = grid && !a.pixel_major ? (a.Ho + R - 1) / R : (a.pixel_major ? a.Ho : 1);
  a.n_ct = grid && !a.pixel_major ? (a.Wo + Wt - 1) / Wt : (a.pixel_major ? a.Wo : 1);
  a.m_tiles = a.n_bt * a.n_rt * a.n_ct;
  return NI * R * Wt;
}

template <int BM, int NP>
static int launch_split_cfg(FwdArgs& a, hipStream_t stream) {
  constexpr int BN = 64;
  constexpr int lds = split_lds_bytes<BN, BM, NP>();
  static_assert(lds <= 160 * 1024, "LDS budget of one CU");
  auto kern = fused_split_kernel<BN, BM, NP>;
  static bool flags[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "fused forward (split): hipGetDevice failed");
  if (!flags[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return set_error(BT_ERR_HIP_BASE, "fused forward (split): cannot raise the dynamic LDS limit");
    flags[dev] = true;
  }
  char nm[160];
  snprintf(nm, sizeof(nm), "fused_split_kernel<%d,%d,bf16x%d,%d terms>", BN, BM, NP, NP == 3 ? 6 : 3);
  note_kernel(nm);
  hipLaunchKernelGGL(kern, dim3((unsigned)a.total_blocks), dim3(512), lds, stream, a);
  return check_launch("fused forward (split)");
}

// Returns BT_OK when the launch was taken, 1 when this flavour does not apply (the caller runs the fp32 kernels), < 0 on error.
int launch_split(FwdArgs& a, hipStream_t stream) {
  const int mode = contraction_mode();
  if (mode == 1) return 1;
  // Reparameterization, on-chip draws, packed parameters, whole channel octets, at most 9 taps, 32-bit byte offsets
  if (!a.mu_pk || (((uintptr_t)a.mu_pk | (uintptr_t)a.sig_pk) & 15u) || (a.Cig & 7) || a.T > 9 || a.ep_pool || a.w_elems >= (1ll << 29) ||
      a.x_elems >= (1ll << 29))
    return 1;
  const int Mdom = a.pixel_major ? a.B : a.M;
  if (Mdom < 256) return 1;
  a.n_tiles = (a.Cog + 63) / 64;
  FwdArgs b512 = a, b256 = a;
  const int live512 = Mdom >= 512 ? split_geometry<512>(b512) : 0;
  const int live256 = split_geometry<256>(b256);
  const long long t512 = live512 ? (long long)a.G * a.n_tiles * a.S * b512.m_tiles : 0;
  int bm = 0;
  if (live512 * 100 >= 512 * 85 && t512 >= 256) bm = 512;       // the wide tile when it is filled and the grid covers the chip
  else if (live256 * 100 >= 256 * 85) bm = 256;
  if (!bm) return 1;
  a = bm == 512 ? b512 : b256;
  const long long total = (long long)a.G * a.n_tiles * a.S * a.m_tiles;
  if (total <= 0 || total > 0x7FFFFFFFll) return 1;
  a.total_blocks = (int)total;
  a.kl_slices = total < 256 ? (int)total : 256;
  if (bm == 512) return mode == 2 ? launch_split_cfg<512, 2>(a, stream) : launch_split_cfg<512, 3>(a, stream);
  return mode == 2 ? launch_split_cfg<256, 2>(a, stream) : launch_split_cfg<256, 3>(a, stream);
}

}  // namespace bt

// Contraction arithmetic of the fused forwards (process-wide knob; also env BT_CONTRACTION = f32 | bf16x3 | bf16x2):
// 0 automatic -- exact bf16x3 split (6 product terms, fp32 accumulate) on the bf16 matrix pipe wherever the launch is eligible,
// 1 fp32 MFMA everywhere (the bit-exact fp32 FMA chain), 2 bf16x2 split (3 terms; relative error ~1e-5, opt-in).
extern "C" int bt_set_contraction(int mode) {
  if (mode < 0 || mode > 2) return bt::set_error(BT_ERR_BAD_ARG, "bt_set_contraction: mode must be 0 (auto), 1 (f32) or 2 (bf16x2)");
  bt::g_contraction.store(mode, std::memory_order_relaxed);
  return BT_OK;
}
