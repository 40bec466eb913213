// Host side of the split-precision flavour (bt_fused_split.h): eligibility, tile geometry, launch.
#include <atomic>
#include <stdlib.h>
#include <string.h>

#include "bt_fused_split_quad.h"
#include "bt_fused_split_direct.h"
#include "bt_fused_split_skinny.h"
#include "bt_fused_split_host.h"

namespace bt {

// 0: automatic (6-term exact split where the launch is eligible), 1: fp32 MFMA only, 2: 3-term split (opt-in, ~1e-5 relative)
static std::atomic<int> g_contraction{-1};
int contraction_mode() {
  int m = g_contraction.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("BT_CONTRACTION");
    m = !e ? 0 : (!strcmp(e, "f32") ? 1 : !strcmp(e, "bf16x2") ? 2 : 0);
    g_contraction.store(m, std::memory_order_relaxed);
  }
  return m;
}

template <int BM, int NP, int NPW, int XM, int BN = 64>
static int launch_split_cfg(FwdArgs& a, hipStream_t stream) {
  constexpr int lds = split_lds_bytes<BN, BM, NP>();
  static_assert(lds <= 160 * 1024, "LDS budget of one CU");
  auto kern = fused_split_kernel<BN, BM, NP, NPW, XM>;
  static bool flags[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "fused forward (split): hipGetDevice failed");
  if (!flags[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return set_error(BT_ERR_HIP_BASE, "fused forward (split): cannot raise the dynamic LDS limit");
    flags[dev] = true;
  }
  char nm[160];
  snprintf(nm, sizeof(nm), "fused_split_kernel<%d,%d,bf16x%d,%d terms,npw=%d,xm=%d>", BN, BM, NP, NP == 3 ? 6 : 3, NPW, XM);
  note_kernel(nm);
  split_fill_inverses(a);
  hipLaunchKernelGGL(kern, dim3((unsigned)a.total_blocks), dim3(256 + 64 * NPW), lds, stream, a);
  return check_launch("fused forward (split)");
}

template <int BM, int NPW>
static int launch_split_xm(FwdArgs& a, int mode, int xm, hipStream_t stream) {
  if (mode == 2) return launch_split_cfg<BM, 2, NPW, 0>(a, stream);   // (the opt-in 3-term form keeps the generic fetch)
  if constexpr (BM == 128) {
    if (a.bn32) {   // 32-channel tiles of a launch that would leave CUs idle (launch_split_one); same K order, same bits
      if (xm == 1) return launch_split_cfg<BM, 3, NPW, 1, 32>(a, stream);
      if (xm == 2) return launch_split_cfg<BM, 3, NPW, 2, 32>(a, stream);
      return launch_split_cfg<BM, 3, NPW, 0, 32>(a, stream);
    }
    if (xm == 1) return launch_split_cfg<BM, 3, NPW, 1>(a, stream);
    if (xm == 2) return launch_split_cfg<BM, 3, NPW, 2>(a, stream);
  } else {
    if (xm == 3) return launch_split_cfg<BM, 3, NPW, 3>(a, stream);
    if (xm == 4) return launch_split_cfg<BM, 3, NPW, 4>(a, stream);
    if constexpr (BM == 256) {
      if (xm == 2) return launch_split_cfg<BM, 3, 8, 2>(a, stream);
    }
  }
  return launch_split_cfg<BM, 3, NPW, 0>(a, stream);
}

// Layers with <= 4 input channels per group (the ResNet stems): bt_fused_split_quad.h. Whole-image 512-wide tiles, output through
// the LDS-staged read-out (optionally with the fused 3x3 / stride-2 max-pool, power-of-two pooled widths).
template <bool POOL>
static int launch_quad_cfg(FwdArgs& a, int mode, hipStream_t stream) {
  constexpr int lds = split_lds_bytes<64, 512, 3>();
  auto launch = [&](auto kern, const char* nm) -> int {
    static bool flags[2][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "fused forward (split): hipGetDevice failed");
    if (!flags[mode == 2][dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return set_error(BT_ERR_HIP_BASE, "fused forward (split): cannot raise the dynamic LDS limit");
      flags[mode == 2][dev] = true;
    }
    note_kernel(nm);
    split_fill_inverses(a);
    hipLaunchKernelGGL(kern, dim3((unsigned)a.total_blocks), dim3(512), lds, stream, a);
    return check_launch("fused forward (split, quad)");
  };
  // (the opt-in two-piece form is not instantiated for the stems: they run the exact split in every split mode)
  return launch(fused_split_quad_kernel<3, POOL>, POOL ? "fused_split_quad_kernel<64,512,bf16x3,6 terms,pool=1>" : "fused_split_quad_kernel<64,512,bf16x3,6 terms,pool=0>");
}

static int launch_quad(FwdArgs& a, int mode, hipStream_t stream) {
  if (a.pixel_major || a.T > 64 || !a.out_vec4 || a.HoWo < 2 || a.Wo > 512) return 1;
  int nh, nw, dys, dxs;
  tap_window(a.KH, a.DH, a.SH, a.PH, a.H, a.Ho, false, &nh, &dys);
  tap_window(a.KW, a.DW, a.SW, a.PW, a.W, a.Wo, false, &nw, &dxs);
  constexpr long long XCAP = kQuadXBytes / 24;
  const long long PWt = (long long)(a.Wo - 1) * (dxs ? a.SW : 1) + dxs + 1;
  int NI, R;
  long long tiles_per_sample;
  if (a.HoWo <= 512) {   // whole images
    const long long PHt = (long long)(a.Ho - 1) * (dys ? a.SH : 1) + dys + 1;
    NI = 512 / a.HoWo, R = a.Ho;
    if (NI > a.B) NI = a.B;
    while (NI > 1 && NI * PHt * PWt > XCAP) --NI;
    if (NI * PHt * PWt > XCAP) return 1;
    tiles_per_sample = (a.B + NI - 1) / NI;
  } else {               // a band of whole rows of one image (ImageNet stems: 112 x 112 outputs -> 4 rows per tile)
    if (a.ep_pool) return 1;   // (the pooled read-out needs whole images: the caller pools in a separate pass)
    NI = 1, R = 512 / a.Wo;
    while (R > 1 && ((long long)(R - 1) * (dys ? a.SH : 1) + dys + 1) * PWt > XCAP) --R;
    if (((long long)(R - 1) * (dys ? a.SH : 1) + dys + 1) * PWt > XCAP) return 1;
    tiles_per_sample = (long long)a.B * ((a.Ho + R - 1) / R);
  }
  if ((double)a.M / ((double)tiles_per_sample * 512) < 0.75) return 1;   // the wide tile must be filled
  if (a.ep_pool) {
    const int Wp = a.ep_Wp;
    if ((Wp & (Wp - 1)) != 0 || Wp < 4 || Wp > 16 || a.ep_res) return 1;
  }
  a.n_tiles = (a.Cog + 63) / 64;
  a.t_NI = NI, a.t_R = R, a.t_Wt = a.Wo, a.n_bt = (a.B + NI - 1) / NI, a.n_rt = (a.Ho + R - 1) / R, a.n_ct = 1, a.m_tiles = a.n_bt * a.n_rt;
  const long long total = (long long)a.G * a.n_tiles * a.S * a.m_tiles;
  if (total <= 0 || total > 0x7FFFFFFFll) return 1;
  a.total_blocks = (int)total;
  a.kl_slices = total < 256 ? (int)total : 256;
  return a.ep_pool ? launch_quad_cfg<true>(a, mode, stream) : launch_quad_cfg<false>(a, mode, stream);
}

// 1x1 / stride-1 convolutions with K <= 256 and many pixels (the bottleneck ResNets' expanding / reducing layers): the persistent
// kernel of bt_fused_split_direct.h. Eligibility is geometric (never a matter of S or of the launch split), and its arithmetic is the
// general kernel's, so a layer's results do not depend on which of the two serves it.
static std::atomic<int> g_bn32{-2};   // -2: read BT_BN32 once; -1 automatic; 0 / 1 forced (bt_debug_force_bn32)
static std::atomic<int> g_direct_off{0};
static int launch_direct(FwdArgs& a, hipStream_t stream) {
  if (g_direct_off.load(std::memory_order_relaxed)) return 1;
  if (a.ep_pool) return 1;
  // a 1x1 kernel without padding (any stride), or any window over a 1x1 image whose ONE live tap sits on the pixel (ResNet18 / CIFAR
  // layer4: 3x3, padding 1, 1x1 maps -- the centre tap; Linear layers are 1x1 kernels over 1x1 images)
  if (a.KH == 1 && a.KW == 1 && a.PH == 0 && a.PW == 0) {
    a.d_tap = 0;
  } else if (a.H == 1 && a.W == 1 && a.Ho == 1 && a.Wo == 1 && a.PH % a.DH == 0 && a.PW % a.DW == 0 && a.PH / a.DH < a.KH && a.PW / a.DW < a.KW) {
    a.d_tap = (a.PH / a.DH) * a.KW + a.PW / a.DW;
  } else {
    return 1;
  }
  const bool resident = a.Cig <= kDirectMaxK;
  // K > 256 streams the weights in chunks that all 8 waves draw and meet at: it needs every wave to own pixels (>= 4096 per sample).
  // CIFAR-sized layers with K = 512 (128 pixels per sample: 2 of 8 waves would multiply, on 2 of the 4 matrix pipes) measured
  // 87 us against the general kernel's 57 (profiles/r03f_layers_cfg3.json): they stay there. K <= 256 wins at any size.
  if ((a.Cig & 63) || (!resident && ((a.Cig % (16 * kDirectChunk)) || a.M < 4096)) || a.M < 64) return 1;
  a.n_tiles = (a.Cog + 63) / 64;
  const long long pairs = (long long)a.G * a.n_tiles * a.S;
  const int nsub = (a.M + 63) / 64;
  // chunks of the pixel range per (group, channel tile, sample): ~1024 workgroups in all (four per CU: the tail of an uneven split
  // is a quarter of a workgroup's work), each with at least 64 sub-tiles (8 per wave) to walk
  // Measured at ResNet50 / b256 / S = 16 (tools/ab_direct_wgs.sh, env BT_DIRECT_WGS): the STREAMED flavour re-draws its weight chunks per
  // 512 pixels whatever the split, so extra workgroups only add prologues and KL slices -- one per CU is best (K = 2048 -> 512 on 7x7:
  // 3260 -> 2805 us; K = 1024 -> 256 on 14x14: 3065 -> 2947); a STRIDED resident layer (the downsamples: half of every fetched line
  // is unused) likes short workgroups that spread its fetches (256 -> 512 stride 2 on 56x56: 8001 -> 7253 us at 4096); everything else
  // stays at four per CU.
  static const int wg_env = [] { const char* e = getenv("BT_DIRECT_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();   // measurement knob
  static const int n_cu = [] { int dev = 0, v = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; return v; }();
  const int wg_target = wg_env ? wg_env : !resident ? n_cu : (a.SH > 1 || a.SW > 1) ? 16 * n_cu : 4 * n_cu;
  long long chunks = (wg_target + pairs - 1) / pairs;
  if (chunks > nsub / 64) chunks = nsub / 64;
  if (chunks < 1) {   // few pixels (CIFAR-sized maps): down to FOUR sub-tiles per workgroup (half its waves), as long as that still adds workgroups the
                      // chip has room for (ResNet18 layer3's downsample: 128 -> 256 workgroups, 45.8 -> 39 us; two sub-tiles per workgroup: 47 us)
    chunks = (256 + pairs - 1) / pairs;
    if (chunks > nsub / 4) chunks = nsub / 4;
    if (chunks < 1) chunks = 1;
  }
  int spc = (int)((nsub + chunks - 1) / chunks);
  if (!resident) spc = (spc + 7) & ~7;   // the 8 waves walk 512-pixel tiles together
  chunks = (nsub + spc - 1) / spc;
  const long long total = pairs * chunks;
  if (total <= 0 || total > 0x7FFFFFFFll) return 1;
  a.m_tiles = (int)chunks, a.t_NI = spc, a.t_R = 1, a.t_Wt = 64, a.n_bt = (int)chunks, a.n_rt = a.n_ct = 1;
  a.total_blocks = (int)total;
  a.kl_slices = total < 256 ? (int)total : 256;
  a.inv_n_tiles = inv_u32(a.n_tiles, total), a.inv_m_tiles = inv_u32(a.m_tiles, total), a.inv_S = inv_u32(a.S, total);
  a.inv_rw = inv_u32(a.HoWo, (long long)a.M + 64 * 8 * 2);   // pixel index -> (image, output position)
  a.inv_wt = inv_u32(a.Wo, a.HoWo);                          // output position -> (row, column): strided layers
  const int lds = direct_lds_bytes(a.Cig);
  auto launch = [&](auto kern, const char* nm, bool* flags, int max_lds) -> int {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "fused forward (split, direct): hipGetDevice failed");
    if (!flags[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
        return set_error(BT_ERR_HIP_BASE, "fused forward (split, direct): cannot raise the dynamic LDS limit");
      flags[dev] = true;
    }
    note_kernel(nm);
    hipLaunchKernelGGL(kern, dim3((unsigned)a.total_blocks), dim3(kDirectThreads), lds, stream, a);
    return check_launch("fused forward (split, direct)");
  };
  static bool fr[64] = {}, fs[64] = {};
  if (resident) return launch(fused_split_direct_kernel<true>, "fused_split_direct_kernel<64,8x64,bf16x3,6 terms,resident W>", fr, direct_lds_bytes(kDirectMaxK));
  return launch(fused_split_direct_kernel<false>, "fused_split_direct_kernel<64,8x64,bf16x3,6 terms,streamed W>", fs, direct_lds_bytes(kDirectMaxK + 1));
}

// Layers whose output map is one pixel and whose batch is small (Linear, CIFAR-sized layer4, the classifier head): the split-K kernel of
// bt_fused_split_skinny.h. Geometry of the slices from the layer alone, so the K order never depends on S or on the launch split.
struct SkinnyPlan { int ks, cpt, kh0, nh, kw0, nw, nsl, n_tiles, m_tiles; long long tiles, scratch; };
static bool skinny_plan(int B, int Ci, int H, int W, int Co, int KH, int KW, int PH, int PW, int DH, int DW, int G, int Ho, int Wo, int S, SkinnyPlan* p) {
  if (Ho != 1 || Wo != 1 || B > 1024 || KH * KW > 64) return false;
  const int Cig = Ci / G, Cog = Co / G;
  if (Cig & 63) return false;
  const int kh0 = (PH + DH - 1) / DH, kw0 = (PW + DW - 1) / DW;              // first tap with kh * DH - PH >= 0
  int kh1 = (PH + H - 1) / DH, kw1 = (PW + W - 1) / DW;                      // last tap with kh * DH - PH <= H - 1
  if (kh1 > KH - 1) kh1 = KH - 1;
  if (kw1 > KW - 1) kw1 = KW - 1;
  if (kh1 < kh0 || kw1 < kw0) return false;
  p->kh0 = kh0, p->nh = kh1 - kh0 + 1, p->kw0 = kw0, p->nw = kw1 - kw0 + 1;
  p->ks = (Cig & 127) ? 64 : 128;
  p->cpt = Cig / p->ks;
  p->nsl = p->nh * p->nw * p->cpt;
  p->n_tiles = (Cog + 63) / 64, p->m_tiles = (B + kSkinnyCols - 1) / kSkinnyCols;
  // Measured on ResNet18 / CIFAR (S = 32, b128, rocprofv3 inside the bench graph, tools/trace_layers.py): the classifier head (4
  // workgroups per sample) 37.6 -> 23.8 us; the 1x1 stride-2 downsample into layer4 (16 per sample) 40.7 -> 43.9 against the direct
  // kernel; layer4's 3x3 layers with one live tap (32 per sample) 53 ... 60 -> 60; layer4.0.conv1 with four live taps (64 per sample,
  // 2048 workgroups) 71 -> 140: past a few slices per sample the slabs' HBM round trip (2 x 32 KB per workgroup) and the second round
  // of workgroups cost more than the shorter chains save. So: narrow heads only. The bound is per SAMPLE -- geometry, not S.
  static const int gate = [] { const char* e = getenv("BT_SKINNY_MAX"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();   // measurement knob
  if ((long long)G * p->m_tiles * p->n_tiles * p->nsl > gate) return false;
  p->tiles = (long long)G * S * p->m_tiles * p->n_tiles;
  if (p->tiles > kSkinnyMaxTiles || p->tiles * p->nsl > 0x7FFFFFFFll) return false;
  p->scratch = p->tiles * p->nsl * (64ll * kSkinnyCols * 4);
  return true;
}
long long skinny_scratch_bytes(const bt_conv2d_geom& g, int S) {
  if (g.B <= 0 || g.Ci <= 0 || g.Co <= 0 || g.groups <= 0 || g.Ci % g.groups || g.Co % g.groups || g.sh <= 0 || g.sw <= 0 || g.dh <= 0 || g.dw <= 0) return 0;
  const int Ho = (g.H + 2 * g.ph - g.dh * (g.kh - 1) - 1) / g.sh + 1, Wo = (g.W + 2 * g.pw - g.dw * (g.kw - 1) - 1) / g.sw + 1;
  SkinnyPlan p;
  return skinny_plan(g.B, g.Ci, g.H, g.W, g.Co, g.kh, g.kw, g.ph, g.pw, g.dh, g.dw, g.groups, Ho, Wo, S, &p) ? p.scratch : 0;
}
static std::atomic<int> g_skinny_off{-1};
static int launch_skinny(FwdArgs& a, hipStream_t stream) {
  int off = g_skinny_off.load(std::memory_order_relaxed);
  if (off < 0) {   // env BT_NO_SKINNY=1: measurement knob (tools/trace_layers.py), like the test hook below
    const char* e = getenv("BT_NO_SKINNY");
    off = (e && *e && *e != '0') ? 1 : 0;
    g_skinny_off.store(off, std::memory_order_relaxed);
  }
  if (off || a.ep_pool || !a.sk_scratch || !a.sk_tickets) return 1;
  SkinnyPlan p;
  if (!skinny_plan(a.B, a.Ci, a.H, a.W, a.Co, a.KH, a.KW, a.PH, a.PW, a.DH, a.DW, a.G, a.Ho, a.Wo, a.S, &p)) return 1;
  if (p.scratch > a.sk_scratch_bytes) return 1;   // the caller brought no (or too little) scratch: the other flavours serve the launch
  if ((((uintptr_t)a.sk_scratch) & 15u)) return 1;
  a.sk_ks = p.ks, a.sk_cpt = p.cpt, a.sk_nsl = p.nsl, a.sk_kh0 = p.kh0, a.sk_nh = p.nh, a.sk_kw0 = p.kw0, a.sk_nw = p.nw;
  a.n_tiles = p.n_tiles, a.m_tiles = p.m_tiles;
  a.t_NI = kSkinnyCols, a.t_R = 1, a.t_Wt = 1, a.n_bt = p.m_tiles, a.n_rt = a.n_ct = 1;
  a.total_blocks = (int)(p.tiles * p.nsl);
  a.kl_slices = a.total_blocks < 256 ? a.total_blocks : 256;   // (spread over every workgroup the sweep lengthened all of them: 50 -> 62 us on ResNet18 layer4)
  a.x_vec = (a.HW == 1 && ((((uintptr_t)a.x) & 15u) == 0) && (a.x_sample_stride & 3) == 0 && (a.Ci & 3) == 0) ? 1 : 0;
  const int lds = skinny_lds_bytes(p.ks);
  static bool flags[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return set_error(BT_ERR_HIP_BASE, "fused forward (split, skinny): hipGetDevice failed");
  if (!flags[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fused_split_skinny_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, skinny_lds_bytes(128)) != hipSuccess)
      return set_error(BT_ERR_HIP_BASE, "fused forward (split, skinny): cannot raise the dynamic LDS limit");
    flags[dev] = true;
  }
  note_kernel(p.ks == 128 ? "fused_split_skinny_kernel<64,4x32,bf16x3,6 terms,split-K 128>" : "fused_split_skinny_kernel<64,4x32,bf16x3,6 terms,split-K 64>");
  hipLaunchKernelGGL(fused_split_skinny_kernel, dim3((unsigned)a.total_blocks), dim3(kSkinnyThreads), lds, stream, a);
  return check_launch("fused forward (split, skinny)");
}

// Returns BT_OK when the launch was taken, 1 when this flavour does not apply (the caller runs the fp32 kernels), < 0 on error.
static int launch_split_one(FwdArgs& a, hipStream_t stream) {
  const int mode = contraction_mode();
  if (mode == 1) return 1;
  // Reparameterization, on-chip draws, packed parameters, 32-bit byte offsets
  if (!a.mu_pk || (((uintptr_t)a.mu_pk | (uintptr_t)a.sig_pk) & 15u) || a.w_elems >= (1ll << 29) || a.x_elems >= (1ll << 29)) return 1;
  if (a.Cig <= 4) return launch_quad(a, mode, stream);   // the stems
  // whole channel octets, at most 9 taps, no fused pooling
  if ((a.Cig & 7) || a.ep_pool) return 1;
  if (mode != 2) {   // (the flavours with a single live tap per slice / layer take any window size)
    FwdArgs k = a;
    const int rck = launch_skinny(k, stream);
    if (rck <= 0) { a = k; return rck; }
    FwdArgs d = a;
    const int rcd = launch_direct(d, stream);
    if (rcd <= 0) { a = d; return rcd; }
  }
  if (a.T > 9) return 1;
  const int Mdom = a.pixel_major ? a.B : a.M;
  if (Mdom < 112) return 1;
  a.n_tiles = (a.Cog + 63) / 64;
  FwdArgs b512 = a, b256 = a, b128 = a;
  const int live512 = Mdom >= 512 ? split_geometry<512>(b512) : 0;
  const int live256 = Mdom >= 256 ? split_geometry<256>(b256) : 0;
  const int live128 = split_geometry<128>(b128);
  const long long per = (long long)a.G * a.n_tiles * a.S;
  // Cost of a launch in column-equivalents: rounds of 256 workgroups x (the tile's columns, but never less than the weight
  // synthesis of a stage costs the producers, + the prologue / output stage of a workgroup). Tiles that waste more than a
  // quarter of their columns are not considered.
  auto cost = [&](int live, int BM, const FwdArgs& b) -> double {
    if (!live) return 1e30;
    const double eff = (double)a.M / ((double)b.m_tiles * BM);
    if (eff < 0.75) return 1e30;
    const double rounds = (double)((per * b.m_tiles + 255) / 256);
    const int synth = BM == 128 ? 128 : 256;   // the producers' floor: the 128-wide tile has 8 producer waves, the others 4
    return rounds * ((BM > synth ? BM : synth) + 96);
  };
  const double c512 = cost(live512, 512, b512), c256 = cost(live256, 256, b256), c128 = cost(live128, 128, b128);
  int bm = 0;
  static const int force_bm = [] { const char* e = getenv("BT_FORCE_BM"); return e ? atoi(e) : 0; }();   // measurement knob: prefer this tile width where it is eligible
  if (force_bm == 512 && c512 < 1e30) bm = 512;
  else if (force_bm == 256 && c256 < 1e30) bm = 256;
  else if (force_bm == 128 && c128 < 1e30) bm = 128;
  else if (c512 < 1e30 && c512 <= c256 && c512 <= c128) bm = 512;
  else if (c256 < 1e30 && c256 <= c128) bm = 256;
  else if (c128 < 1e30) bm = 128;
  if (!bm) return 1;
  a = bm == 512 ? b512 : bm == 256 ? b256 : b128;
  // A 128-wide launch that offers at most one workgroup per two CUs (a training step's single sample; an MLP's wide first layer at 8
  // samples) runs in 32-channel tiles instead: twice the workgroups, each drawing half the weights -- the chain of a workgroup of such a
  // layer IS its weight synthesis. The K order does not depend on the channel tile, so the results are the same bits (and the choice may
  // depend on S). BT_BN32 = 0 | 1 forces it off / on where eligible (tests, measurement).
  int bn32_env = g_bn32.load(std::memory_order_relaxed);
  if (bn32_env == -2) {
    const char* e = getenv("BT_BN32");
    bn32_env = e ? (atoi(e) ? 1 : 0) : -1;
    g_bn32.store(bn32_env, std::memory_order_relaxed);
  }
  a.bn32 = 0;
  if (bm == 128 && mode != 2 && a.Cog > 32) {
    const long long wgs64 = per * a.m_tiles;
    a.bn32 = bn32_env >= 0 ? (bn32_env ? 1 : 0) : (2 * wgs64 <= 256 ? 1 : 0);
  }
  long long per_t = per;
  if (a.bn32) a.n_tiles = (a.Cog + 31) / 32, per_t = (long long)a.G * a.n_tiles * a.S;
  const long long total = per_t * a.m_tiles;
  if (total <= 0 || total > 0x7FFFFFFFll) return 1;
  a.total_blocks = (int)total;
  a.kl_slices = total < 256 ? (int)total : 256;
  // x fetch mode (bt_fused_split.h): tiny input planes are read as 16-byte vectors
  const bool xal = (((uintptr_t)a.x) & 15u) == 0 && (a.x_sample_stride & 3) == 0;
  int xm = 0;
  if (xal && a.HW == 1) xm = 1;
  else if (xal && bm != 128 && !a.pixel_major && a.HW > 1 && (a.W & 3) == 0 && a.t_Wt == a.Wo && split_row_mode(a)) xm = split_row_mode(a);
  else if (xal && bm != 128 && !a.pixel_major && a.HW > 1 && split_plane_flat(a)) xm = 3, a.x_flat = 1;
  else if (xal && bm == 128 && !a.pixel_major && a.H == 2 && a.W == 2 && split_plane_flat(a)) xm = 2;   // whole 2x2 planes (a strided 3x3 down to 1x1 maps)
  else if (xal && bm != 512 && a.row_taps && a.H == 2 && a.W == 2 && a.KW == 3 && a.PW == 1 && a.SW == 1 && a.DW == 1) xm = 2;   // a row tile's patch is the whole 2x2 plane
  else if (xal && a.pixel_major && a.H == 2 && a.W == 2 && a.KH == 3 && a.KW == 3 && a.PH == 1 && a.PW == 1 && a.SH == 1 && a.SW == 1 && a.DH == 1 && a.DW == 1) xm = 2;
  if (bm == 512) return launch_split_xm<512, 4>(a, mode, xm, stream);
  if (bm == 256) return launch_split_xm<256, 8>(a, mode, xm, stream);
  return launch_split_xm<128, 8>(a, mode, xm, stream);
}

// Pixel-major tiles (2..4-pixel outputs) prune the padding taps per pixel: the first choice. When their patch does not fit (a
// stride-2 3x3 from 4x4 to 2x2 maps: up to 9 input pixels per output pixel and image), tiles of whole images -- every active
// tap once for all pixels -- usually do, and beat the fp32 kernels (165 -> 104 us on ResNet18's layer3.0.conv1).
int launch_split(FwdArgs& a, hipStream_t stream) {
  FwdArgs t = a;
  // Two-row maps with a stride-1 window (ResNet18 / CIFAR layer3: 3x3 on 2x2): tiles of (images x ONE output row) before the
  // pixel-major ones. A row's pixels share 6 of the 9 taps: 3 MFMA steps and 6 weight draws per octet for 2 pixels instead of
  // 2 x (2 steps, 4 draws), and half the workgroups. The choice is geometric (never a matter of S or of the tile width), so
  // the tap pairing -- the K order -- of a layer stays the same for every launch split.
  if (a.pixel_major && a.Ho == 2 && a.Wo >= 2 && a.SH == 1 && a.KH == 3 && a.PH == 1 && a.DH == 1 && !a.ep_pool) {
    t.pixel_major = 0, t.out_vec4 = 0, t.row_taps = 1;
    const int rcr = launch_split_one(t, stream);
    if (rcr <= 0) { a = t; return rcr; }
    t = a;
  }
  int rc = launch_split_one(t, stream);
  if (rc == 1 && a.pixel_major) {
    t = a;
    t.pixel_major = 0;
    t.out_vec4 = 0;   // 2..4-pixel rows: the scalar output stage
    rc = launch_split_one(t, stream);
  }
  if (rc <= 0) a = t;   // the plan that ran (bt_last_launch_info reads it)
  return rc;
}

}  // namespace bt

// Test hook (not part of include/bt_hip.h): 1 keeps 1x1 convolutions off the direct kernel, so a test can compare the two flavours.
extern "C" void bt_debug_disable_direct(int off) { bt::g_direct_off.store(off ? 1 : 0, std::memory_order_relaxed); }
extern "C" void bt_debug_force_bn32(int v) { bt::g_bn32.store(v < 0 ? -1 : (v ? 1 : 0), std::memory_order_relaxed); }
extern "C" void bt_debug_disable_skinny(int off) { bt::g_skinny_off.store(off ? 1 : 0, std::memory_order_relaxed); }

// Contraction arithmetic of the fused forwards (process-wide knob; also env BT_CONTRACTION = f32 | bf16x3 | bf16x2):
// 0 automatic -- exact bf16x3 split (6 product terms, fp32 accumulate) on the bf16 matrix pipe wherever the launch is eligible,
// 1 fp32 MFMA everywhere (the bit-exact fp32 FMA chain), 2 bf16x2 split (3 terms; relative error ~1e-5, opt-in).
extern "C" int bt_set_contraction(int mode) {
  if (mode < 0 || mode > 2) return bt::set_error(BT_ERR_BAD_ARG, "bt_set_contraction: mode must be 0 (auto), 1 (f32) or 2 (bf16x2)");
  bt::g_contraction.store(mode, std::memory_order_relaxed);
  return BT_OK;
}
