// K1-K5: the fused stochastic forward (+KL) of the four Bayesian layers as ONE kernel template.
//
//   D_s[co][m] = sum_k W_s[co][k] * X_s[k][m]      m = (b, ho, wo),  k = (ci, kh, kw)
//
//   W_s = mu + log1p(exp(rho)) * eps_s   is synthesised per K-tile straight into LDS (never in HBM):
//         coalesced float4 loads of (mu, rho) [+ injected eps | on-chip Philox + Box-Muller],
//         two roundings exactly like the reference (tmp = sigma*eps; w = mu + tmp);
//   X_s   is the implicit-GEMM (im2col) view of the NCHW input, gathered per K-tile through a small
//         k -> (offset, dh, dw) table; for Linear it is the row-major activation tile;
//   the contraction runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain,
//         rtol 1e-4 with K up to 4608 rules out bf16 inputs and gfx950 has no xf32);
//   Flipout keeps TWO accumulators over one staged x tile: mean path (mu, x) and perturbation path
//         (sigma*eps, x o s_in), s_in/s_out applied in registers;
//   KL    is accumulated by the blocks that own (m-tile 0, sample 0) while (mu, rho) are in registers,
//         reduced with wave shuffles and finished in fixed order by the last-arriving block;
//   MC    samples are a grid dimension: one launch computes S samples, (mu, rho) re-reads hit L2.
//
// Replaces the ATen chains at reference layers/variational_layers/linear_variational.py:163-181,
// conv_variational.py:366-385, flipout_layers/linear_flipout.py:149-174, conv_flipout.py:376-417.
#pragma once
#include "bt_api_internal.h"

namespace bt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FwdArgs {
  const float *x, *mu_w, *rho_w, *mu_b, *rho_b, *pmu_w, *psig_w, *pmu_b, *psig_b;
  const float *eps_w, *eps_b, *sign_in, *sign_out;
  float* out;
  float* kl_out;
  double* slots;
  unsigned* counter;
  long long x_sample_stride, x_elems, out_elems, w_elems;
  int B, Ci, H, W, Co, KH, KW, SH, SW, PH, PW, DH, DW, G;
  int Ho, Wo, HoWo, M, K, Cig, Cog, S;
  int n_tiles, m_tiles, total_blocks;
  int w_vec, x_vec;  // float4 paths allowed (K % 4 == 0 and 16-B aligned bases)
  int do_kl;
  uint32_t seed_lo, seed_hi, call, layer_id, sample0;
  const uint32_t* call_base;  // device word added to `call` (fresh draws on graph replay), or null
  const float *ep_scale, *ep_shift, *ep_res;  // fused output stage (bt_epilogue)
  long long ep_res_stride;
  int ep_relu;
};

// Blocks are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2). Give every XCD a
// CONTIGUOUS range of the logical block order, which is n-tile-major: an XCD then works on few
// n-tiles for all samples and m-tiles, so its (mu, rho) working set stays in its own L2.
// Bijective for any grid size (cdna_hip_programming.md, T1).
__device__ __forceinline__ int xcd_remap(int orig, int n) {
  const int q = n >> 3, r = n & 7, xcd = orig & 7, i = orig >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

template <int BN, int BM, int WAVES_N, bool FLIP, bool LINEAR, bool TRANS>
__global__ __launch_bounds__(256) void fused_fwd_kernel(const FwdArgs a) {
  constexpr int BK = 32;
  constexpr int KQ = BK / 4;                  // float4 quads per tile row
  constexpr int RPP = 256 / KQ;               // tile rows covered per loader pass (32)
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WTN = BN / WAVES_N, WTM = BM / WAVES_M;
  constexpr int TN = WTN / 32, TM = WTM / 32;
  constexpr int WS = BN + 1, XS = BM + 1;     // odd strides: conflict-free transposed ds_write_b32
  constexpr int NW = FLIP ? 2 : 1;
  static_assert(TN >= 1 && TM >= 1 && BN % RPP == 0 && BM % 32 == 0, "tile shape");
  static_assert(!LINEAR || TRANS, "Linear always stores with lanes along the output features");

  // one LDS array (a second __shared__ object next to staging arrays can de-pipeline the k-loop)
  constexpr int W_WORDS = BK * WS, X_WORDS = BK * XS;
  constexpr int TAB_WORDS = 2 * BK * 4;
  __shared__ __attribute__((aligned(16))) float smem[NW * (W_WORDS + X_WORDS) + TAB_WORDS + 16];
  float* const Wt0 = smem;
  float* const Wt1 = smem + W_WORDS;                       // FLIP: sigma*eps tile
  float* const Xt0 = smem + NW * W_WORDS;
  float* const Xt1 = Xt0 + X_WORDS;                        // FLIP: x o s_in tile
  int4* const ktab = reinterpret_cast<int4*>(smem + NW * (W_WORDS + X_WORDS));
  double* const red = reinterpret_cast<double*>(smem + NW * (W_WORDS + X_WORDS) + TAB_WORDS);  // 4 doubles + flag
  int* const flag = reinterpret_cast<int*>(red + 4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wn = wave / WAVES_M, wm = wave % WAVES_M;

  int L = xcd_remap(blockIdx.x, a.total_blocks);
  const int mt = L % a.m_tiles;
  L /= a.m_tiles;
  const int s = L % a.S;
  L /= a.S;
  const int nt = L % a.n_tiles;
  const int g = L / a.n_tiles;
  const int n0 = nt * BN, m0 = mt * BM;
  const bool kl_block = a.do_kl && mt == 0 && s == 0;
  const uint32_t sample = a.sample0 + (uint32_t)s;
  const int K = a.K;

  RngKey key_w;
  key_w.seed_lo = a.seed_lo;
  key_w.seed_hi = a.seed_hi;
  key_w.call = a.call + (a.call_base ? __builtin_nontemporal_load(a.call_base) : 0u);
  key_w.layer_tensor = layer_tensor_word(a.layer_id, 0);
  uint32_t skey_in = 0, skey_out = 0;
  if (FLIP) {
    RngKey ks = key_w;
    ks.layer_tensor = layer_tensor_word(a.layer_id, 2);
    if (!a.sign_in) skey_in = sign_stream_key(ks, sample);
    ks.layer_tensor = layer_tensor_word(a.layer_id, 3);
    if (!a.sign_out) skey_out = sign_stream_key(ks, sample);
  }

  // ---- loader constants -----------------------------------------------------------------------
  const int kq = tid % KQ, lr0 = tid / KQ;  // W loader (and Linear x loader): row lr0 + 32*pass, k quad kq
  const float* const xs = a.x + (long long)s * a.x_sample_stride;
  const float* const eps_w_s = a.eps_w ? a.eps_w + (long long)s * a.w_elems : nullptr;
  const float* const sin_s = (FLIP && a.sign_in) ? a.sign_in + (long long)s * a.x_elems : nullptr;

  // conv gather: this thread owns column xm of the x tile for k rows xk0, xk0 + XKP, ...
  constexpr int XKP = 256 / BM > 0 ? 256 / BM : 1;
  const int xm = tid % BM, xk0 = tid / BM;
  int hi0 = 0, wi0 = 0;
  long long xoff0 = 0;
  bool mvalid = false;
  if (!LINEAR) {
    const int m = m0 + xm;
    mvalid = (m < a.M) && (tid < BM * XKP);
    const int mm = mvalid ? m : 0;
    const int b = mm / a.HoWo, p = mm - b * a.HoWo;
    const int ho = p / a.Wo, wo = p - ho * a.Wo;
    hi0 = ho * a.SH - a.PH;
    wi0 = wo * a.SW - a.PW;
    xoff0 = ((long long)b * a.Ci + (long long)g * a.Cig) * a.H * a.W + (long long)hi0 * a.W + wi0;
  }

  auto fill_ktab = [&](int buf, int k0) {
    if (!LINEAR && tid < BK) {
      const int k = k0 + tid;
      int4 e = make_int4(0, 1 << 24, 1 << 24, 0);  // fails the bounds test
      if (k < K) {
        const int taps = a.KH * a.KW;
        const int ci = k / taps, r = k - ci * taps;
        const int kh = r / a.KW, kw = r - kh * a.KW;
        e.x = ci * a.H * a.W + kh * a.DH * a.W + kw * a.DW;
        e.y = kh * a.DH;
        e.z = kw * a.DW;
      }
      ktab[buf * BK + tid] = e;
    }
  };

  f32x16 acc[NW][TN][TM];
#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[w][i][j][r] = 0.f;

  double kl_acc = 0.0;
  fill_ktab(0, 0);
  __syncthreads();

  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
    // ---------------- W tile: (mu, rho) [+eps] -> registers -> sampled weights -> LDS (transposed) ------
#pragma unroll
    for (int p = 0; p < BN / RPP; ++p) {
      const int r = lr0 + p * RPP;
      const int co_g = n0 + r;
      const int k = k0 + 4 * kq;
      const bool rv = co_g < a.Cog;
      const long long widx = ((long long)g * a.Cog + co_g) * K + k;
      float mu[4] = {0.f, 0.f, 0.f, 0.f}, rho[4] = {0.f, 0.f, 0.f, 0.f}, ep[4] = {0.f, 0.f, 0.f, 0.f};
      float pm[4] = {0.f, 0.f, 0.f, 0.f}, ps[4] = {1.f, 1.f, 1.f, 1.f};
      bool ev[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ev[j] = rv && (k + j < K);
      if (a.w_vec) {
        if (ev[0]) {  // K % 4 == 0: the quad is all-in or all-out
          const float4 m4 = *reinterpret_cast<const float4*>(a.mu_w + widx);
          const float4 r4 = *reinterpret_cast<const float4*>(a.rho_w + widx);
          mu[0] = m4.x, mu[1] = m4.y, mu[2] = m4.z, mu[3] = m4.w;
          rho[0] = r4.x, rho[1] = r4.y, rho[2] = r4.z, rho[3] = r4.w;
          if (eps_w_s) {
            const float4 e4 = *reinterpret_cast<const float4*>(eps_w_s + widx);
            ep[0] = e4.x, ep[1] = e4.y, ep[2] = e4.z, ep[3] = e4.w;
          } else {
            philox_normal4(key_w, sample, (uint32_t)(widx >> 2), ep);
          }
          if (kl_block) {
            const float4 a4 = *reinterpret_cast<const float4*>(a.pmu_w + widx);
            const float4 b4 = *reinterpret_cast<const float4*>(a.psig_w + widx);
            pm[0] = a4.x, pm[1] = a4.y, pm[2] = a4.z, pm[3] = a4.w;
            ps[0] = b4.x, ps[1] = b4.y, ps[2] = b4.z, ps[3] = b4.w;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ev[j]) {
            mu[j] = a.mu_w[widx + j];
            rho[j] = a.rho_w[widx + j];
            if (eps_w_s) {
              ep[j] = eps_w_s[widx + j];
            } else {
              float z[4];
              philox_normal4(key_w, sample, (uint32_t)((widx + j) >> 2), z);
              const int sel = (int)((widx + j) & 3);
              ep[j] = sel == 0 ? z[0] : sel == 1 ? z[1] : sel == 2 ? z[2] : z[3];
            }
            if (kl_block) {
              pm[j] = a.pmu_w[widx + j];
              ps[j] = a.psig_w[widx + j];
            }
          }
        }
      }
      float kl4 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sg = softplus(rho[j]);
        const float dl = __fmul_rn(sg, ep[j]);
        float w0, w1 = 0.f;
        if (FLIP) {
          w0 = mu[j];
          w1 = dl;
        } else {
          w0 = __fadd_rn(mu[j], dl);
        }
        Wt0[(4 * kq + j) * WS + r] = ev[j] ? w0 : 0.f;
        if (FLIP) Wt1[(4 * kq + j) * WS + r] = ev[j] ? w1 : 0.f;
        if (kl_block) kl4 += ev[j] ? kl_term(mu[j], sg, pm[j], ps[j]) : 0.f;
      }
      if (kl_block) kl_acc += (double)kl4;
    }

    // ---------------- x tile -----------------------------------------------------------------------------
    if (LINEAR) {
#pragma unroll
      for (int p = 0; p < BM / RPP; ++p) {
        const int r = lr0 + p * RPP;
        const int m = m0 + r;
        const int k = k0 + 4 * kq;
        const long long xo = (long long)m * K + k;
        float v[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {1.f, 1.f, 1.f, 1.f};
        const bool rv = m < a.M;
        if (a.x_vec) {
          if (rv && k < K) {
            const float4 x4 = *reinterpret_cast<const float4*>(xs + xo);
            v[0] = x4.x, v[1] = x4.y, v[2] = x4.z, v[3] = x4.w;
            if (FLIP && sin_s) {
              const float4 s4 = *reinterpret_cast<const float4*>(sin_s + xo);
              sg[0] = s4.x, sg[1] = s4.y, sg[2] = s4.z, sg[3] = s4.w;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (rv && k + j < K) {
              v[j] = xs[xo + j];
              if (FLIP && sin_s) sg[j] = sin_s[xo + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Xt0[(4 * kq + j) * XS + r] = v[j];
          if (FLIP) {
            const float sj = sin_s ? sg[j] : hash_sign(skey_in, (uint32_t)(xo + j));
            Xt1[(4 * kq + j) * XS + r] = __fmul_rn(v[j], sj);
          }
        }
      }
    } else {
      float v[BK / XKP], sg[BK / XKP];
      long long xo[BK / XKP];
#pragma unroll
      for (int p = 0; p < BK / XKP; ++p) {
        const int kk = xk0 + p * XKP;
        const int4 e = ktab[buf * BK + kk];
        const bool ok = mvalid && (unsigned)(hi0 + e.y) < (unsigned)a.H && (unsigned)(wi0 + e.z) < (unsigned)a.W;
        xo[p] = xoff0 + e.x;
        v[p] = ok ? xs[xo[p]] : 0.f;
        if (FLIP) sg[p] = (ok && sin_s) ? sin_s[xo[p]] : 1.f;
      }
#pragma unroll
      for (int p = 0; p < BK / XKP; ++p) {
        const int kk = xk0 + p * XKP;
        if (tid < BM * XKP) {
          Xt0[kk * XS + xm] = v[p];
          if (FLIP) {
            const float sj = sin_s ? sg[p] : hash_sign(skey_in, (uint32_t)xo[p]);
            Xt1[kk * XS + xm] = __fmul_rn(v[p], sj);
          }
        }
      }
      fill_ktab(buf ^ 1, k0 + BK);
    }
    __syncthreads();

    // ---------------- contraction on the fp32 matrix cores ----------------------------------------------
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float af[NW][TN], bf[NW][TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        af[0][i] = Wt0[(kk + lh) * WS + wn * WTN + i * 32 + li];
        if (FLIP) af[NW - 1][i] = Wt1[(kk + lh) * WS + wn * WTN + i * 32 + li];
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        bf[0][j] = Xt0[(kk + lh) * XS + wm * WTM + j * 32 + li];
        if (FLIP) bf[NW - 1][j] = Xt1[(kk + lh) * XS + wm * WTM + j * 32 + li];
      }
#pragma unroll
      for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[w][i][j] = TRANS ? __builtin_amdgcn_mfma_f32_32x32x2f32(bf[w][j], af[w][i], acc[w][i][j], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i], bf[w][j], acc[w][i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---------------- epilogue: bias draw, Flipout sign_out, store ------------------------------------------
  float* const bias0 = Wt0;        // reparam: mu_b + sigma_b*eps_b ; flipout: mu_b
  float* const bias1 = Wt0 + BN;   // flipout: sigma_b*eps_b
  if (tid < BN) {
    float b0 = 0.f, b1 = 0.f;
    const int co_g = n0 + tid;
    if (a.mu_b && co_g < a.Cog) {
      const int co = g * a.Cog + co_g;
      float e;
      if (a.eps_b) {
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
      if (FLIP) {
        b0 = a.mu_b[co];
        b1 = dl;
      } else {
        b0 = __fadd_rn(a.mu_b[co], dl);
      }
    }
    bias0[tid] = b0;
    if (FLIP) bias1[tid] = b1;
  }
  __syncthreads();

  float* const out_s = a.out + (long long)s * a.out_elems;
  const float* const sout_s = (FLIP && a.sign_out) ? a.sign_out + (long long)s * a.out_elems : nullptr;
  const float* const res_s = a.ep_res ? a.ep_res + (long long)s * a.ep_res_stride : nullptr;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    int b_col = 0, p_col = 0;
    if (!TRANS) {  // lanes run along m: decode this lane's column once
      const int m = m0 + wm * WTM + j * 32 + li;
      b_col = m / a.HoWo;
      p_col = m - b_col * a.HoWo;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        int co_l, m;
        long long oidx;
        if (TRANS) {  // D[m][co]: lanes along co, HoWo == 1
          co_l = wn * WTN + i * 32 + li;
          m = m0 + wm * WTM + j * 32 + row;
          oidx = (long long)m * a.Co + (long long)g * a.Cog + n0 + co_l;
        } else {      // D[co][m]: lanes along the spatial index (NCHW-contiguous)
          co_l = wn * WTN + i * 32 + row;
          m = m0 + wm * WTM + j * 32 + li;
          oidx = ((long long)b_col * a.Co + (long long)g * a.Cog + n0 + co_l) * a.HoWo + p_col;
        }
        if (n0 + co_l < a.Cog && m < a.M) {
          float v = __fadd_rn(acc[0][i][j][r], bias0[co_l]);
          if (FLIP) {
            const float so = sout_s ? sout_s[oidx] : hash_sign(skey_out, (uint32_t)oidx);
            v = __fadd_rn(v, __fmul_rn(__fadd_rn(acc[NW - 1][i][j][r], bias1[co_l]), so));
          }
          if (a.ep_scale) {
            const int co = g * a.Cog + n0 + co_l;
            v = __fadd_rn(__fmul_rn(v, a.ep_scale[co]), a.ep_shift[co]);
          }
          if (res_s) v = __fadd_rn(v, res_s[oidx]);
          if (a.ep_relu) v = fmaxf(v, 0.f);
          out_s[oidx] = v;
        }
      }
    }
  }

  // ---------------- KL finish ------------------------------------------------------------------------------
  if (!kl_block) return;
  const double bsum = block_sum_256(kl_acc, red);
  if (tid == 0) *flag = publish_and_ticket(a.slots, a.counter, g * a.n_tiles + nt, bsum, (unsigned)(a.G * a.n_tiles)) ? 1 : 0;
  __syncthreads();
  if (!*flag) return;
  double bacc = 0.0;
  if (a.mu_b)
    for (int c = tid; c < a.Co; c += 256) bacc += (double)kl_term(a.mu_b[c], softplus(a.rho_b[c]), a.pmu_b[c], a.psig_b[c]);
  __syncthreads();
  const double bias_sum = block_sum_256(bacc, red);
  if (tid == 0) {
    double wsum = 0.0;
    const int nslots = a.G * a.n_tiles;
    for (int i = 0; i < nslots; ++i) wsum += __hip_atomic_load(&a.slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float kl = (float)(wsum / (double)a.w_elems);
    if (a.mu_b) kl += (float)(bias_sum / (double)a.Co);
    a.kl_out[0] = kl;
    __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// host side ------------------------------------------------------------------------------------------------


}  // namespace bt
