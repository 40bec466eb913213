// Tile selection + launch for one FLIP flavour (included by bt_fused_reparam.hip / bt_fused_flipout.hip
// so the two sets of instantiations compile in parallel).
#pragma once
#include "bt_fused_fwd.h"

namespace bt {

template <int BN, int BM, int WAVES_N, bool FLIP, bool LINEAR, bool TRANS>
static int launch_cfg(FwdArgs& a, hipStream_t stream) {
  a.n_tiles = (a.Cog + BN - 1) / BN;
  a.m_tiles = (a.M + BM - 1) / BM;
  const long long total = (long long)a.G * a.n_tiles * a.S * a.m_tiles;
  if (total <= 0 || total > 0x7FFFFFFFll) return set_error(BT_ERR_UNSUPPORTED, "fused forward: grid too large");
  if (a.do_kl && a.G * a.n_tiles > kMaxSlots) return set_error(BT_ERR_UNSUPPORTED, "fused forward: too many KL slots");
  a.total_blocks = (int)total;
  hipLaunchKernelGGL((fused_fwd_kernel<BN, BM, WAVES_N, FLIP, LINEAR, TRANS>), dim3((unsigned)total), dim3(256), 0, stream, a);
  return check_launch("fused forward");
}

template <bool FLIP, bool LINEAR, bool TRANS>
static int pick_tile(FwdArgs& a, hipStream_t stream) {
  // block tile = BN output channels x BM output positions, 4 waves of 32x32 MFMA tiles.
  if (a.M <= 32) return launch_cfg<128, 32, 4, FLIP, LINEAR, TRANS>(a, stream);
  if (a.Cog <= 32) return launch_cfg<32, 128, 1, FLIP, LINEAR, TRANS>(a, stream);
  if (a.M <= 64) return launch_cfg<64, 64, 2, FLIP, LINEAR, TRANS>(a, stream);
  if (a.Cog <= 64) return launch_cfg<64, 128, 2, FLIP, LINEAR, TRANS>(a, stream);
  return launch_cfg<128, 128, 2, FLIP, LINEAR, TRANS>(a, stream);
}

template <bool FLIP>
static int launch_flavour(bool linear, FwdArgs& a, hipStream_t stream) {
  if (linear) return pick_tile<FLIP, true, true>(a, stream);
  if (a.HoWo == 1) return pick_tile<FLIP, false, true>(a, stream);
  return pick_tile<FLIP, false, false>(a, stream);
}

}  // namespace bt
