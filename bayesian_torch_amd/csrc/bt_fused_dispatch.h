// Tile selection + launch for one FLIP flavour (included by bt_fused_reparam.hip / bt_fused_flipout.hip
// so the two sets of instantiations compile in parallel).
#pragma once
#include <stdlib.h>

#include "bt_fused_fast.h"

namespace bt {

// Tile geometry of the specialised kernel (bt_fused_fast.h): t_NI images x t_R output rows x t_Wt output columns per tile,
// chosen so the x patch of 4 channels (worst case: every tap active) fits the LDS x buffer. Returns false when this launch
// has to run the general kernel.
template <int BM, bool LINEAR, bool FLIP = false>
static bool fast_geometry(FwdArgs& a) {
  static const bool forced_off = getenv("BT_FORCE_GENERIC") != nullptr;  // A/B hook for tests and benchmarks
  if (forced_off || !a.mu_pk || (((uintptr_t)a.mu_pk | (uintptr_t)a.sig_pk) & 15u) || a.T > kMaxTaps / 2 || a.w_elems >= (1ll << 29) ||
      a.x_elems >= (1ll << 29))
    return false;
  constexpr long long XW = x_words<BM, FLIP>();
  const int dys = (a.KH - 1) * a.DH, dxs = (a.KW - 1) * a.DW;
  auto fits = [&](int NI, int R, int Wt) {
    const long long PHt = (long long)(R - 1) * (dys ? a.SH : 1) + dys + 1, PWt = (long long)(Wt - 1) * (dxs ? a.SW : 1) + dxs + 1;
    const long long PCH = NI * PHt * PWt;
    return 4 * PCH <= XW && PCH < 65536;
  };
  int NI, R, Wt;
  if (LINEAR || a.HoWo == 1 || a.pixel_major) {
    NI = BM, R = 1, Wt = 1;               // one output position per image: tile = BM images (pixel-major: of one pixel)
    if (!LINEAR && !fits(NI, R, Wt)) return false;
  } else if (a.HoWo <= BM) {
    NI = BM / a.HoWo, R = a.Ho, Wt = a.Wo;  // whole images
    while (NI > 1 && !fits(NI, R, Wt)) --NI;
    if (!fits(NI, R, Wt)) return false;
  } else if (a.Wo <= BM) {
    NI = 1, R = BM / a.Wo, Wt = a.Wo;       // a band of rows of one image
    while (R > 1 && !fits(NI, R, Wt)) --R;
    if (!fits(NI, R, Wt)) return false;
  } else {
    NI = 1, R = 1, Wt = BM;                 // a segment of one row
    if (!fits(NI, R, Wt)) return false;
  }
  const bool grid = a.pixel_major || (!LINEAR && a.HoWo > 1);
  // Row-chunk staging of the x patch (16-byte pieces of input rows copied straight into LDS): needs 16-byte aligned rows
  // and the slightly wider patch to fit with the same tile. Flipout stages x through registers (it multiplies by the signs).
  static const bool no_cvec = getenv("BT_NO_XCVEC") != nullptr;  // A/B hook
  a.x_cvec = (!no_cvec && !LINEAR && (((uintptr_t)a.x) & 15u) == 0 && (a.x_sample_stride & 3) == 0 && (((long long)a.Ci * a.HW) & 3) == 0 && (a.Cig & 3) == 0) ? 1 : 0;
  a.x_rows = 0;
  static const bool no_rows = getenv("BT_NO_XROWS") != nullptr;  // A/B hook
  if (!no_rows && !LINEAR && !FLIP && !a.pixel_major && a.HoWo > 1 && (a.W & 3) == 0 && (((uintptr_t)a.x) & 15u) == 0 && (a.x_sample_stride & 3) == 0) {
    const long long PHt = (long long)(R - 1) * (dys ? a.SH : 1) + dys + 1;
    int xa, n;
    if (Wt == a.Wo) {
      n = row_chunks(-a.PW, -a.PW + (Wt - 1) * a.SW + dxs, a.W, &xa);  // the tile spans the row: exact (all taps active is the worst case)
    } else {
      n = (((Wt - 1) * a.SW + dxs + 1 + 3) >> 2) + 1;                    // column segments: any alignment of the first column
    }
    const long long PCH = NI * PHt * 4 * n + 4;
    if (4 * PCH <= XW && PCH < 65536) a.x_rows = 1;
  }
  a.t_NI = NI, a.t_R = R, a.t_Wt = Wt;
  a.n_bt = (a.B + NI - 1) / NI;
  a.n_rt = grid ? (a.Ho + R - 1) / R : 1;
  a.n_ct = grid ? (a.Wo + Wt - 1) / Wt : 1;
  a.m_tiles = a.n_bt * a.n_rt * a.n_ct;
  return true;
}

template <typename Kern>
static int ensure_lds(Kern kern, int lds, bool* flags) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "fused forward: hipGetDevice failed");
  if (!flags[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return set_error(BT_ERR_HIP_BASE, "fused forward: cannot raise the dynamic LDS limit");
    flags[dev] = true;
  }
  return BT_OK;
}

template <int BN, int BM, int CWN, bool FLIP, bool LINEAR, bool TRANS, int XMODE, bool POOL = false>
static int launch_fast(const FwdArgs& a, hipStream_t stream) {
  constexpr int lds = fused_lds_bytes<BN, BM, FLIP>();
  // narrow conv tiles: 8 producer waves (their accumulators leave room for 12 waves of <= 168 registers)
  constexpr int NPW = (!LINEAR && BM <= 128 && BN * BM <= 128 * 128) ? 8 : 4;
  auto fk = fused_fast_kernel<BN, BM, CWN, FLIP, LINEAR, TRANS, false, XMODE, NPW, POOL>;
  static bool fflags[64] = {};
  if (int rc = ensure_lds(fk, lds, fflags)) return rc;
  {
    char nm[160];
    snprintf(nm, sizeof(nm), "fused_fast_kernel<%d,%d,%d,%s,%s,%s,inj=0,xmode=%d,npw=%d,pool=%d>", BN, BM, CWN, FLIP ? "flip" : "reparam",
             LINEAR ? "linear" : "conv", TRANS ? "trans" : "notrans", XMODE, NPW, POOL ? 1 : 0);
    note_kernel(nm);
  }
  hipLaunchKernelGGL(fk, dim3((unsigned)a.total_blocks), dim3(256 + 64 * NPW), lds, stream, a);
  return check_launch("fused forward (fast)");
}

template <int BN, int BM, int CWN, bool FLIP, bool LINEAR, bool TRANS, bool INJ>
static int launch_cfg(FwdArgs& a, hipStream_t stream) {
  constexpr int lds = fused_lds_bytes<BN, BM, FLIP>();
  static_assert(lds <= 160 * 1024, "LDS budget of one CU");
  a.n_tiles = (a.Cog + BN - 1) / BN;
  bool fast = false;
  if constexpr (!INJ) fast = fast_geometry<BM, LINEAR, FLIP>(a);  // injected draws are the parity/debug mode: always the general kernel
  if (a.ep_pool && !(fast && TRANS && !FLIP && BM >= 128 && a.x_rows && a.out_vec4 && !a.pixel_major && a.t_R == a.Ho && a.t_Wt == a.Wo))
    return set_error(BT_ERR_UNSUPPORTED, "fused max-pool: this launch's tiles do not hold whole output images");
  if (!fast) {  // general kernel: BM consecutive (b, ho, wo), or pixel-major
    if (a.pixel_major) {
      a.mt_per_pixel = (a.B + BM - 1) / BM;
      a.m_tiles = a.HoWo * a.mt_per_pixel;
    } else {
      a.mt_per_pixel = 1;
      a.m_tiles = (a.M + BM - 1) / BM;
    }
    a.patch_ok = (!LINEAR && (a.pixel_major || (a.HoWo <= BM && BM % a.HoWo == 0))) ? 1 : 0;
  }
  const long long total = (long long)a.G * a.n_tiles * a.S * a.m_tiles;
  if (total <= 0 || total > 0x7FFFFFFFll) return set_error(BT_ERR_UNSUPPORTED, "fused forward: grid too large");
  a.total_blocks = (int)total;
  a.kl_slices = total < 256 ? (int)total : 256;  // workgroups that sweep a slice of the weights for KL (4 wave slots each)
  if constexpr (!INJ) {
    if (fast) {
      // x staging mode (bt_fused_fast.h): row chunks need the wide spatial tiles, channel vectors the narrow ones
      constexpr bool has_rows = !LINEAR && !FLIP && BM >= 128, has_cvec = !LINEAR && BM <= 128;
      if constexpr (has_rows) {
        if (a.x_rows) {
          if constexpr (TRANS) {
            if (a.ep_pool) return launch_fast<BN, BM, CWN, FLIP, LINEAR, TRANS, 1, true>(a, stream);
          }
          return launch_fast<BN, BM, CWN, FLIP, LINEAR, TRANS, 1>(a, stream);
        }
      }
      if constexpr (has_cvec) {
        if (a.x_cvec && (a.HW == 1 || a.HW == 4)) return launch_fast<BN, BM, CWN, FLIP, LINEAR, TRANS, 2>(a, stream);
      }
      return launch_fast<BN, BM, CWN, FLIP, LINEAR, TRANS, 0>(a, stream);
    }
  }
  if constexpr (BM <= (FLIP ? 128 : 256)) {
    auto kern = fused_fwd_kernel<BN, BM, CWN, FLIP, LINEAR, TRANS, INJ>;
    static bool gflags[64] = {};
    if (int rc = ensure_lds(kern, lds, gflags)) return rc;
    {
      char nm[160];
      snprintf(nm, sizeof(nm), "fused_fwd_kernel<%d,%d,%d,%s,%s,%s,inj=%d>", BN, BM, CWN, FLIP ? "flip" : "reparam", LINEAR ? "linear" : "conv",
               TRANS ? "trans" : "notrans", INJ ? 1 : 0);
      note_kernel(nm);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(kThreads), lds, stream, a);
    return check_launch("fused forward");
  } else {
    return set_error(BT_ERR_UNSUPPORTED, "fused forward: this tile exists in the fast flavour only");
  }
}

static inline long long tiles_for(const FwdArgs& a, int BN, int BM) {
  const long long nt = (a.Cog + BN - 1) / BN;
  long long mt = a.pixel_major ? (long long)a.HoWo * ((a.B + BM - 1) / BM) : (a.M + BM - 1) / BM;
  if (!a.pixel_major && a.HoWo > 1 && a.HoWo <= BM) mt = (a.B + BM / a.HoWo - 1) / (BM / a.HoWo);  // whole-image tiles
  return (long long)a.G * nt * a.S * mt;
}

template <bool FLIP, bool LINEAR, bool TRANS, bool INJ>
static int pick_tile_by_size(FwdArgs& a, hipStream_t stream);

template <bool FLIP, bool LINEAR, bool TRANS, bool INJ>
static int pick_tile(FwdArgs& a, hipStream_t stream) {
  const int rc = pick_tile_by_size<FLIP, LINEAR, TRANS, INJ>(a, stream);
  if constexpr (!LINEAR && !INJ) {
    // The fused max-pool needs tiles of whole images. When the size-driven choice has none (small batches pick narrow
    // tiles), take the narrowest tile that holds an image; launch_cfg launches nothing when it declines.
    if (rc == BT_ERR_UNSUPPORTED && a.ep_pool) {
      if constexpr (!FLIP) {  // (the pooled read-out lives in the row-chunk instantiations: Reparameterization, aligned x)
        if (a.HoWo <= 128) return launch_cfg<64, 128, 2, FLIP, LINEAR, TRANS, INJ>(a, stream);
        if (a.HoWo <= 256) return launch_cfg<64, 256, 1, FLIP, LINEAR, TRANS, INJ>(a, stream);
        if (a.HoWo <= 512) return launch_cfg<64, 512, 1, FLIP, LINEAR, TRANS, INJ>(a, stream);
      }
    }
  }
  return rc;
}

template <bool FLIP, bool LINEAR, bool TRANS, bool INJ>
static int pick_tile_by_size(FwdArgs& a, hipStream_t stream) {
  // Workgroup tile = BN output channels x BM output positions; 4 consumer waves of (BN/CWN) x (BM/CWM) each.
  // Wide BM amortises one weight draw over more MFMA work (the producers' VALU budget); a launch should still
  // offer >= 256 workgroups (one per CU), so tiles shrink when the grid would not fill the chip.
  constexpr long long kCUs = 256;
  const int Mdom = a.pixel_major ? a.B : a.M;
  if (Mdom <= 32) return launch_cfg<128, 32, 4, FLIP, LINEAR, TRANS, INJ>(a, stream);
  if (Mdom <= 64) return launch_cfg<64, 64, 2, FLIP, LINEAR, TRANS, INJ>(a, stream);
  if (a.Cog <= 32) return launch_cfg<32, 128, 1, FLIP, LINEAR, TRANS, INJ>(a, stream);
  if constexpr (!FLIP) {  // wide tiles: one accumulator set fits in the consumers' registers (Flipout carries two)
    if constexpr (!LINEAR && !INJ) {  // 512-wide: fast flavour only (x as a patch); halves the weight-synthesis work per MFMA
      if (Mdom >= 512 && tiles_for(a, 64, 512) >= kCUs) {
        FwdArgs probe = a;
        // only when the wide tile is actually filled (a 256-pixel image whose 2-image patch does not fit would leave half of it dead)
        if (fast_geometry<512, false>(probe) && probe.t_NI * probe.t_R * probe.t_Wt >= 448)
          return launch_cfg<64, 512, 1, FLIP, LINEAR, TRANS, INJ>(a, stream);
      }
    }
    if (Mdom >= 256 && a.Cog > 64 && tiles_for(a, 128, 256) >= kCUs) return launch_cfg<128, 256, 2, FLIP, LINEAR, TRANS, INJ>(a, stream);
    if (Mdom >= 256 && tiles_for(a, 64, 256) >= kCUs) return launch_cfg<64, 256, 1, FLIP, LINEAR, TRANS, INJ>(a, stream);
  }
  if constexpr (FLIP && !LINEAR && !INJ) {
    // Flipout's wide tile: 64x256, fast flavour only (two accumulator sets of 64 registers; x as a patch within the
    // 128-column LDS budget). Halves the weight synthesis per MFMA on the large feature maps.
    if (Mdom >= 256 && ((a.SH == 1 && a.SW == 1) || a.T > 9) && tiles_for(a, 64, 256) >= kCUs) {  // (strided 3x3: measured slower; stems: faster)
      FwdArgs probe = a;
      if (fast_geometry<256, false, true>(probe) && probe.t_NI * probe.t_R * probe.t_Wt >= 224)
        return launch_cfg<64, 256, 1, FLIP, LINEAR, TRANS, INJ>(a, stream);
    }
  }
  if (a.Cog > 64 && tiles_for(a, 128, 128) >= kCUs) return launch_cfg<128, 128, 2, FLIP, LINEAR, TRANS, INJ>(a, stream);
  return launch_cfg<64, 128, 2, FLIP, LINEAR, TRANS, INJ>(a, stream);
}

template <bool FLIP, bool INJ>
static int launch_flavour(bool linear, FwdArgs& a, hipStream_t stream) {
  if (linear && a.w_vec && a.x_vec) return pick_tile<FLIP, true, true, INJ>(a, stream);   // float4 fast path
  if (a.HoWo == 1 || a.pixel_major || a.out_vec4) return pick_tile<FLIP, false, true, INJ>(a, stream);  // incl. any other Linear: a 1x1 conv
  return pick_tile<FLIP, false, false, INJ>(a, stream);
}

}  // namespace bt
