// Host side of the split-precision Flipout forward (bt_fused_split.h, FLIP = true): eligibility, tile geometry, launch.
#include "bt_fused_split_host.h"

namespace bt {

template <int XM>
static int launch_split_flip_cfg(FwdArgs& a, hipStream_t stream) {
  constexpr int BN = 64, BM = 256, NP = 3, NPW = 4;
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
  hipLaunchKernelGGL(kern, dim3((unsigned)a.total_blocks), dim3(256 + 64 * NPW), lds, stream, a);
  return check_launch("fused forward (split, flipout)");
}

// Returns BT_OK when the launch was taken, 1 when this flavour does not apply (the caller runs the fp32 kernels), < 0 on error.
// One tile: 64 channels x 256 output positions of whole images / row bands (the two accumulator sets of Flipout fill the
// consumers' registers at 32 x 128 per wave); the patch of one octet plane has to fit 301 pixels.
int launch_split_flip(FwdArgs& a, hipStream_t stream) {
  if (contraction_mode() != 0) return 1;   // f32: the fp32 kernels; bf16x2: Reparameterization only
  if (!a.mu_pk || (((uintptr_t)a.mu_pk | (uintptr_t)a.sig_pk) & 15u) || a.w_elems >= (1ll << 29) || a.x_elems >= (1ll << 29)) return 1;
  if ((a.Cig & 7) || a.T > 9 || a.ep_pool || a.pixel_major) return 1;
  if (a.M < 256) return 1;
  FwdArgs b = a;
  b.n_tiles = (b.Cog + 63) / 64;
  const int live = split_geometry<256, true>(b);
  if (!live) return 1;
  if ((double)b.M / ((double)b.m_tiles * 256) < 0.75) return 1;   // the tile must be filled
  const long long total = (long long)b.G * b.n_tiles * b.S * b.m_tiles;
  if (total <= 0 || total > 0x7FFFFFFFll) return 1;
  b.total_blocks = (int)total;
  b.kl_slices = total < 256 ? (int)total : 256;
  const bool xal = (((uintptr_t)b.x) & 15u) == 0 && (b.x_sample_stride & 3) == 0;
  const bool rows = xal && b.HW > 1 && b.SH == 1 && b.SW == 1 && (b.W & 3) == 0 && b.t_Wt == b.Wo && split_rows_cover(b);
  const int rc = rows ? launch_split_flip_cfg<3>(b, stream) : launch_split_flip_cfg<0>(b, stream);
  if (rc == BT_OK) a = b;
  return rc;
}

}  // namespace bt
