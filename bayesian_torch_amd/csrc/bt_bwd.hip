// Backward of the four fused forwards: hand-written data-gradient and weight-gradient
// kernels that REGENERATE the draws on chip from the forward's counter coordinates -- no eps / sign / sampled-weight
// tensor is ever written to or read from HBM (injected draws, the parity mode, are read instead) -- and apply the chain
// rule to (mu, rho) in the weight-gradient's own output stage:
//     W_s = mu + softplus(rho) * eps_s        dL/dmu = sum_s dL/dW_s        dL/drho = sigmoid(rho) * sum_s eps_s * dL/dW_s
//     Flipout: out = conv(x, mu) + conv(x o s_in, softplus(rho) * eps) o s_out   -> the mean path feeds mu, the
//     perturbation path (inputs x o s_in, upstream g o s_out) feeds rho.
// Reference arithmetic differentiated: layers/variational_layers/linear_variational.py:163-181, conv_variational.py:366-385,
// flipout_layers/linear_flipout.py:149-174, conv_flipout.py:376-417 (the reference itself relies on torch autograd).
//
// Both kernels are implicit GEMMs on fp32 MFMA (v_mfma_f32_32x32x2_f32: the exact fp32 FMA chain), 64 x 64 output tiles,
// 4 waves (one 32 x 32 sub-tile each), operands staged through LDS 16 reduction steps at a time:
//   dgrad  dX_s[ci][(b,h,w)]  = sum_{co,tap} W_s[co][ci][tap] * g_s[co][(b,ho,wo)]      W_s regenerated per stage
//   wgrad  D_s[co][(tap,ci)]  = sum_{(b,ho,wo)} g_s[co][.] * x_s[ci][tap window]        then  acc_mu += D_s, acc_rho += eps_s o D_s
// wgrad splits the samples over `groups` workgroups per tile (enough to fill the chip), writes one partial per group, and
// a finishing kernel adds the partials in index order (deterministic), multiplies by sigmoid(rho) and un-permutes the
// tap-major draw layout into the parameters' own [Co][Ci/g][kh][kw].
#include "bt_api_internal.h"

namespace bt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct BwdArgs {
  const float *x, *g;               // x [S or 1][B][Ci][H][W], g = dL/dout [S][B][Co][Ho][Wo]
  const float *mu_pk, *sig_pk;      // tap-major packed (mu, softplus(rho)): [Co][T][Cig4]
  const float *rho_w;               // natural layout (finishing kernel: sigmoid)
  const float *eps_w, *sign_in, *sign_out;  // injected draws (natural layouts) or null
  float *dx, *dmu, *drho;           // dx [S][B][Ci][H][W]; dmu / drho natural [Co][Cig][T]
  float *part;                      // wgrad partials [2][groups][Co][T][Cig4]
  float *part_d;                    // dgrad partials [dchunks][S][x_elems] (behind wgrad's: the two passes may run in ONE launch)
  int pair_wx, pair_wy, pair_nw, pair_dx, pair_dy;   // bwd_pair_kernel: wgrad's grid (x, y) and block count, dgrad's grid (x, y)
  int fin_nw;                       // bwd_finish_pair_kernel: blocks of wgrad's finishing pass (the rest are dgrad's)
  long long x_sample_stride, x_elems, out_elems, w_elems;
  int B, Ci, H, W, Co, KH, KW, SH, SW, PH, PW, DH, DW, G;
  int Ho, Wo, Cig, Cog, Cig4, T, S, flip, groups;
  int dchunks, dchunk;              // dgrad: the reduction over output channels in dchunks pieces of dchunk channels (partials in `part`)
  int sgroups, mchunk;              // wgrad: groups = sgroups (sample s handled by s % sgroups) x chunks of mchunk rows of M = B*Ho*Wo
  uint32_t seed_lo, seed_hi, call, layer_id, sample0;
  const uint32_t* call_base;
  // bt_conv2d_bwd_kl: the layer's KL term differentiated in wgrad's finishing pass (all null / 0 otherwise)
  const float *mu_w, *pmu_w, *psig_w, *gkl;
  int kl_laplace;
};

constexpr int kBK = 16;   // reduction steps per LDS stage
constexpr int kLS = 68;   // LDS row stride (floats): 64 + 4 keeps the float4 stores aligned and the row reads conflict-free

__device__ __forceinline__ RngKey bwd_key(const BwdArgs& a, uint32_t tensor) {
  RngKey k;
  k.seed_lo = a.seed_lo, k.seed_hi = a.seed_hi;
  k.call = a.call + (a.call_base ? *a.call_base : 0u);
  k.layer_tensor = layer_tensor_word(a.layer_id, tensor);
  return k;
}

// ------------------------------------------------------------------------------------------------------------ dgrad
// grid: (m tiles of 64 positions of one sample) x (ci tiles of 64 per group) x (S * G * dchunks). Taps that connect no
// position of the tile to an output are skipped (3x3 on 1x1 maps: 8 of 9). With dchunks > 1 a workgroup reduces over its
// piece of the output channels and writes a partial; dgrad_finish_kernel adds the pieces in order.
template <bool FLIP>
__device__ __forceinline__ void dgrad_body(const BwdArgs& a, float (*As)[kBK][kLS], float (*Bs)[kBK][kLS], const int bx, const int by, const int bz) {
  // As [kk][ci]: W (Flipout: mu | sigma*eps); Bs [kk][col]: g (Flipout: g | g o s_out)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int kc = bz / (a.S * a.G), zr = bz - kc * (a.S * a.G);
  const int s = zr / a.G, grp = zr % a.G;
  const int co_lo = kc * a.dchunk, co_hi = co_lo + a.dchunk < a.Cog ? co_lo + a.dchunk : a.Cog;
  const int ci0 = by * 64, m0 = bx * 64;
  const int HW = a.H * a.W, M = a.B * HW, HoWo = a.Ho * a.Wo;
  const uint32_t sample = a.sample0 + (uint32_t)s;
  const RngKey kw = bwd_key(a, 0);
  uint32_t skey_in = 0, skey_out = 0;
  if (FLIP && !a.sign_in) skey_in = sign_stream_key(bwd_key(a, 2), sample), skey_out = sign_stream_key(bwd_key(a, 3), sample);
  const float* const gs = a.g + (long long)s * a.out_elems;
  // this thread's B-stage columns: col = tid & 63 for kk = (tid >> 6) + 4 j
  const int bcol = tid & 63, m = m0 + bcol;
  const bool mok = m < M;
  const int mb = mok ? m / HW : 0, mh = mok ? (m % HW) / a.W : 0, mw = mok ? m % a.W : 0;
  // A stage: thread = (kk = tid >> 4, ci quad = tid & 15)
  const int akk = tid >> 4, acq = tid & 15, aci = ci0 + 4 * acq;

  f32x16 acc[FLIP ? 2 : 1];
#pragma unroll
  for (int w = 0; w < (FLIP ? 2 : 1); ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[w][r] = 0.f;

  for (int tap = 0; tap < a.T; ++tap) {
    const int kh = tap / a.KW, kwi = tap - kh * a.KW;
    // output position feeding input pixel (mh, mw) through this tap: ho = (mh + PH - kh*DH) / SH when divisible and in range
    const int nh = mh + a.PH - kh * a.DH, nw = mw + a.PW - kwi * a.DW;
    const int ho = nh / a.SH, wo = nw / a.SW;
    const bool pok = mok && nh >= 0 && nw >= 0 && nh - ho * a.SH == 0 && nw - wo * a.SW == 0 && ho < a.Ho && wo < a.Wo;
    const long long gpix = pok ? (long long)mb * a.Co * HoWo + (long long)ho * a.Wo + wo : 0;
    if (!__syncthreads_or(pok ? 1 : 0)) continue;  // this tap meets only padding for every position of the tile
    for (int co0 = co_lo; co0 < co_hi; co0 += kBK) {
      __syncthreads();
      {  // ---- A: 4 consecutive input channels of (co, tap) = one Philox block of the forward's stream
        const int cog = co0 + akk, co = grp * a.Cog + cog;
        float4 wmu = make_float4(0, 0, 0, 0), wd = make_float4(0, 0, 0, 0);
        if (cog < co_hi && aci < a.Cig4) {
          const uint32_t e = ((uint32_t)co * (uint32_t)a.T + (uint32_t)tap) * (uint32_t)a.Cig4 + (uint32_t)aci;
          const float4 m4 = *reinterpret_cast<const float4*>(a.mu_pk + e), s4 = *reinterpret_cast<const float4*>(a.sig_pk + e);
          float ep[4];
          if (a.eps_w) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ep[j] = aci + j < a.Cig ? a.eps_w[(long long)s * a.w_elems + ((long long)co * a.Cig + aci + j) * a.T + tap] : 0.f;
          } else {
            philox_normal4(kw, sample, e >> 2, ep);
          }
          wd = make_float4(__fmul_rn(s4.x, ep[0]), __fmul_rn(s4.y, ep[1]), __fmul_rn(s4.z, ep[2]), __fmul_rn(s4.w, ep[3]));
          wmu = FLIP ? m4 : make_float4(__fadd_rn(m4.x, wd.x), __fadd_rn(m4.y, wd.y), __fadd_rn(m4.z, wd.z), __fadd_rn(m4.w, wd.w));
        }
        *reinterpret_cast<float4*>(&As[0][akk][4 * acq]) = wmu;
        if (FLIP) *reinterpret_cast<float4*>(&As[FLIP ? 1 : 0][akk][4 * acq]) = wd;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // ---- B: upstream gradient at the position this tap connects
        const int kk = (tid >> 6) + 4 * j, cog = co0 + kk;
        float v = 0.f, vp = 0.f;
        if (pok && cog < co_hi) {
          const long long oi = gpix + (long long)(grp * a.Cog + cog) * HoWo;
          v = gs[oi];
          if (FLIP) vp = __fmul_rn(v, a.sign_out ? a.sign_out[(long long)s * a.out_elems + oi] : hash_sign(skey_out, (uint32_t)oi));
        }
        Bs[0][kk][bcol] = v;
        if (FLIP) Bs[FLIP ? 1 : 0][kk][bcol] = vp;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kBK / 2; ++q) {
#pragma unroll
        for (int w = 0; w < (FLIP ? 2 : 1); ++w)
          acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[w][2 * q + lh][wr + li], Bs[w][2 * q + lh][wc + li], acc[w], 0, 0, 0);
      }
    }
  }
  // D[row = ci][col = position]: lane holds column wc + li, registers hold rows (r&3) + 8 (r>>2) + 4 lh
  const int mcol = m0 + wc + li;
  if (mcol < M) {
    const int b = mcol / HW, hw = mcol - b * HW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ci < a.Cig) {
        const long long xi = ((long long)b * a.Ci + grp * a.Cig + ci) * HW + hw;
        float v = acc[0][r];
        if (FLIP) v = __fadd_rn(v, __fmul_rn(acc[FLIP ? 1 : 0][r], a.sign_in ? a.sign_in[(long long)s * a.x_elems + xi] : hash_sign(skey_in, (uint32_t)xi)));
        if (a.dchunks > 1) a.part_d[((long long)kc * a.S + s) * a.x_elems + xi] = v;
        else a.dx[(long long)s * a.x_elems + xi] = v;
      }
    }
  }
}

template <bool FLIP>
__global__ __launch_bounds__(256) void dgrad_kernel(const BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float As[FLIP ? 2 : 1][kBK][kLS];
  __shared__ __attribute__((aligned(16))) float Bs[FLIP ? 2 : 1][kBK][kLS];
  dgrad_body<FLIP>(a, As, Bs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// dx <- sum of the dchunks partials, in piece order (block bid of nb)
__device__ __forceinline__ void dgrad_finish_body(const BwdArgs& a, const int bid, const int nb) {
  const long long n = (long long)a.S * a.x_elems;
  for (long long i = (long long)bid * 256 + threadIdx.x; i < n; i += (long long)nb * 256) {
    float v = 0.f;
    for (int c = 0; c < a.dchunks; ++c) v = __fadd_rn(v, a.part_d[(long long)c * n + i]);
    a.dx[i] = v;
  }
}
__global__ __launch_bounds__(256) void dgrad_finish_kernel(const BwdArgs a) { dgrad_finish_body(a, blockIdx.x, gridDim.x); }

// n / d (0 <= n < 2^31, d >= 1) through the reciprocal inv = ceil(2^32 / d): the estimate is the quotient or one more, one compare fixes it.
__device__ __forceinline__ uint32_t recip32(int d) { return d > 1 ? 0xFFFFFFFFu / (uint32_t)d + 1u : 0u; }
__device__ __forceinline__ int div_recip(int n, int d, uint32_t inv) {
  if (d == 1) return n;
  uint32_t q = __umulhi((uint32_t)n, inv);
  if (q * (uint32_t)d > (uint32_t)n) --q;
  return (int)q;
}

// ------------------------------------------------------------------------------------------------------------ wgrad
// grid: (k' tiles of 64 over T*Cig4) x (co tiles of 64 per group) x (groups * G); group q = (sample group q % sgroups: samples
// s = q % sgroups + i * sgroups) x (chunk q / sgroups of the reduction axis M = B*Ho*Wo: rows [c * mchunk, (c + 1) * mchunk))
template <bool FLIP>
__device__ __forceinline__ void wgrad_body(const BwdArgs& a, float (*As)[kBK][kLS], float (*Bs)[kBK][kLS], const int bx, const int by, const int bz) {
  // As [mm][co]: g (Flipout: g | g o s_out); Bs [mm][k']: x window (Flipout: x | x o s_in)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
  const int sgrp = bz / a.G, grp = bz % a.G;
  const int co0 = by * 64, k0 = bx * 64;
  const int KP = a.T * a.Cig4, HW = a.H * a.W, HoWo = a.Ho * a.Wo, M = a.B * HoWo;
  const RngKey kw = bwd_key(a, 0);
  // this thread's staging columns: col = tid & 63 (a co for A, a k' for B), mm = (tid >> 6) + 4 j
  const int scol = tid & 63;
  const int kp = k0 + scol, tap = kp < KP ? kp / a.Cig4 : 0, kci = kp - tap * a.Cig4;
  const bool kok = kp < KP && kci < a.Cig;
  const int kh = tap / a.KW, kwi = tap - kh * a.KW;
  const int aco = co0 + scol;
  const bool cok = aco < a.Cog;

  f32x16 acc_mu, acc_rho;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_mu[r] = 0.f, acc_rho[r] = 0.f;

  const uint32_t inv_howo = recip32(HoWo), inv_wo = recip32(a.Wo);
  const int sg0 = sgrp % a.sgroups, mlo = (sgrp / a.sgroups) * a.mchunk, mhi = mlo + a.mchunk < M ? mlo + a.mchunk : M;
  for (int s = sg0; s < a.S; s += a.sgroups) {
    const uint32_t sample = a.sample0 + (uint32_t)s;
    uint32_t skey_in = 0, skey_out = 0;
    if (FLIP && !a.sign_in) skey_in = sign_stream_key(bwd_key(a, 2), sample), skey_out = sign_stream_key(bwd_key(a, 3), sample);
    const float* const gs = a.g + (long long)s * a.out_elems;
    const float* const xs = a.x + (long long)s * a.x_sample_stride;
    f32x16 d[FLIP ? 2 : 1];
#pragma unroll
    for (int w = 0; w < (FLIP ? 2 : 1); ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) d[w][r] = 0.f;
    for (int mm0 = mlo; mm0 < mhi; mm0 += kBK) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mm = (tid >> 6) + 4 * j, m = mm0 + mm;
        const bool mok = m < mhi;
        // (reciprocal multiplies: two integer divisions per element and stage were ~a quarter of this loop's instructions)
        const int b = mok ? div_recip(m, HoWo, inv_howo) : 0, p = mok ? m - b * HoWo : 0, ho = div_recip(p, a.Wo, inv_wo), wo = p - ho * a.Wo;
        float ga = 0.f, gp = 0.f, xv = 0.f, xp = 0.f;
        if (mok && cok) {
          const long long oi = ((long long)b * a.Co + grp * a.Cog + aco) * HoWo + p;
          ga = gs[oi];
          if (FLIP) gp = __fmul_rn(ga, a.sign_out ? a.sign_out[(long long)s * a.out_elems + oi] : hash_sign(skey_out, (uint32_t)oi));
        }
        const int hi = ho * a.SH - a.PH + kh * a.DH, wi = wo * a.SW - a.PW + kwi * a.DW;
        if (mok && kok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) {
          const long long xi = ((long long)b * a.Ci + grp * a.Cig + kci) * HW + (long long)hi * a.W + wi;
          xv = xs[xi];
          if (FLIP) xp = __fmul_rn(xv, a.sign_in ? a.sign_in[(long long)s * a.x_elems + xi] : hash_sign(skey_in, (uint32_t)xi));
        }
        As[0][mm][scol] = ga, Bs[0][mm][scol] = xv;
        if (FLIP) As[FLIP ? 1 : 0][mm][scol] = gp, Bs[FLIP ? 1 : 0][mm][scol] = xp;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kBK / 2; ++q) {
#pragma unroll
        for (int w = 0; w < (FLIP ? 2 : 1); ++w)
          d[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[w][2 * q + lh][wr + li], Bs[w][2 * q + lh][wc + li], d[w], 0, 0, 0);
      }
    }
    // chain rule of this sample: the mean path adds to dmu, the eps-weighted (perturbation) path to the rho accumulator
    const int kpl = k0 + wc + li, ltap = kpl < KP ? kpl / a.Cig4 : 0, lci = kpl - ltap * a.Cig4;
    // On-chip draws: the four lanes of a quad (columns 4 q .. 4 q + 3 = the four channels of ONE Philox block of a row) used to run the
    // same block each, once per accumulator row -- 16 blocks per lane and sample. Now lane j of the quad draws the blocks of the rows
    // r = j (mod 4) and the quad exchanges them: 4 blocks per lane, the same values.
    float Z[4][4];
    if (!a.eps_w) {
      const bool quad_ok = (kpl & ~3) < KP && (lci & ~3) < a.Cig;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = 4 * t + (li & 3), cog = co0 + wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
        Z[t][0] = Z[t][1] = Z[t][2] = Z[t][3] = 0.f;
        if (quad_ok && cog < a.Cog)
          philox_normal4(kw, sample, (((uint32_t)(grp * a.Cog + cog) * (uint32_t)a.T + (uint32_t)ltap) * (uint32_t)a.Cig4 + (uint32_t)lci) >> 2, Z[t]);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cog = co0 + wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const bool ok = kpl < KP && lci < a.Cig && cog < a.Cog;
      float e = 0.f;
      if (a.eps_w) {
        if (ok) e = a.eps_w[(long long)s * a.w_elems + ((long long)(grp * a.Cog + cog) * a.Cig + lci) * a.T + ltap];
      } else {
        const int src = (lane & ~3) | (r & 3);   // the quad lane that drew row r's block
        const float v0 = __shfl(Z[r >> 2][0], src, 64), v1 = __shfl(Z[r >> 2][1], src, 64), v2 = __shfl(Z[r >> 2][2], src, 64), v3 = __shfl(Z[r >> 2][3], src, 64);
        const int sel = lci & 3;
        e = sel == 0 ? v0 : sel == 1 ? v1 : sel == 2 ? v2 : v3;
        if (!ok) e = 0.f;
      }
      acc_mu[r] = __fadd_rn(acc_mu[r], d[0][r]);
      acc_rho[r] = __fadd_rn(acc_rho[r], __fmul_rn(e, d[FLIP ? 1 : 0][r]));
    }
  }
  // partials [2][groups][Co][KP]
  const int kpl = k0 + wc + li;
  if (kpl < KP) {
    const long long plane = (long long)a.Co * KP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cog = co0 + wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (cog < a.Cog) {
        const long long o = ((long long)sgrp * a.Co + grp * a.Cog + cog) * KP + kpl;
        a.part[o] = acc_mu[r];
        a.part[(long long)a.groups * plane + o] = acc_rho[r];
      }
    }
  }
}

template <bool FLIP>
__global__ __launch_bounds__(256) void wgrad_kernel(const BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float As[FLIP ? 2 : 1][kBK][kLS];
  __shared__ __attribute__((aligned(16))) float Bs[FLIP ? 2 : 1][kBK][kLS];
  wgrad_body<FLIP>(a, As, Bs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Both passes of a layer in ONE launch: they read the same upstream gradient and are independent of each other, and at one sample per
// step each is a few hundred short workgroups -- 28 + 35 us back to back on ResNet18 / CIFAR, the longer of the two side by side.
// Blocks [0, pair_nw) are wgrad's grid (x fastest, then y, then z) -- the longer pass first --, the rest dgrad's.
template <bool FLIP>
__global__ __launch_bounds__(256) void bwd_pair_kernel(const BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float As[FLIP ? 2 : 1][kBK][kLS];
  __shared__ __attribute__((aligned(16))) float Bs[FLIP ? 2 : 1][kBK][kLS];
  int L = blockIdx.x;
  if (L < a.pair_nw) {
    const int q = L / a.pair_wx, bx = L - q * a.pair_wx, bz = q / a.pair_wy, by = q - bz * a.pair_wy;
    wgrad_body<FLIP>(a, As, Bs, bx, by, bz);
  } else {
    L -= a.pair_nw;
    const int q = L / a.pair_dx, bx = L - q * a.pair_dx, bz = q / a.pair_dy, by = q - bz * a.pair_dy;
    dgrad_body<FLIP>(a, As, Bs, bx, by, bz);
  }
}

// (dKL/dmu, dKL/drho) of ONE element of kl_div's mean (base_variational_layer.py:70-72; 'laplace': :74-97), times gs = upstream / n.
// The op order of kl_normal_bwd_kernel below (which calls it): a gradient built here and one built there are the same bits.
__device__ __forceinline__ void kl_elem_grad(float m, float r, const float* pmu, const float* psig, long long i, int laplace, float gs, float& dm, float& dr) {
  const float sq = softplus(r);
  const float sg = __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __builtin_amdgcn_exp2f(__fmul_rn(-1.4426950408889634f, r))));
  float gm, gq;
  if (laplace) {  // d/dmu E|w| = erf(mu / (sq sqrt 2)); d/dsq = sqrt(2/pi) exp(-mu^2 / (2 sq^2)) - 1/sq
    const float z = __fmul_rn(m, __builtin_amdgcn_rcpf(__fmul_rn(sq, 1.4142135623730951f)));
    gm = erff(z);
    gq = __fsub_rn(__fmul_rn(0.7978845608028654f, __builtin_amdgcn_exp2f(__fmul_rn(-1.4426950408889634f, __fmul_rn(z, z)))), __builtin_amdgcn_rcpf(sq));
  } else {
    const float ip = __builtin_amdgcn_rcpf(__fmul_rn(psig[i], psig[i]));
    gm = __fmul_rn(__fsub_rn(m, pmu[i]), ip);
    gq = __fsub_rn(__fmul_rn(sq, ip), __builtin_amdgcn_rcpf(sq));
  }
  dm = __fmul_rn(gm, gs);
  dr = __fmul_rn(__fmul_rn(gq, sg), gs);
}

// natural element (co, ci, tap) <- sum over the groups' partials at the tap-major position, in group order; with a.gkl also the
// layer's KL term: grad = (contraction path) + (KL path), the sum autograd's AccumulateGrad would have made of the two tensors.
__device__ __forceinline__ void wgrad_finish_body(const BwdArgs& a, const int bid, const int nb) {
  const long long n = (long long)a.Co * a.Cig * a.T;
  const int KP = a.T * a.Cig4;
  const long long plane = (long long)a.Co * KP;
  const float gkl_s = a.gkl ? a.gkl[0] / (float)n : 0.f;
  for (long long i = (long long)bid * 256 + threadIdx.x; i < n; i += (long long)nb * 256) {
    const int tap = (int)(i % a.T);
    const long long rc = i / a.T;
    const int ci = (int)(rc % a.Cig), co = (int)(rc / a.Cig);
    const long long o = (long long)co * KP + (long long)tap * a.Cig4 + ci;
    float sm = 0.f, sr = 0.f;
    for (int gq = 0; gq < a.groups; ++gq) sm = __fadd_rn(sm, a.part[gq * plane + o]), sr = __fadd_rn(sr, a.part[(a.groups + gq) * plane + o]);
    const float r = a.rho_w[i];
    const float sg = __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __builtin_amdgcn_exp2f(__fmul_rn(-1.4426950408889634f, r))));  // sigmoid(rho)
    float om = sm, orr = __fmul_rn(sr, sg);
    if (a.gkl) {
      float km, kr;
      kl_elem_grad(a.mu_w[i], r, a.pmu_w, a.psig_w, i, a.kl_laplace, gkl_s, km, kr);
      om = __fadd_rn(om, km), orr = __fadd_rn(orr, kr);
    }
    if (a.dmu) a.dmu[i] = om;
    if (a.drho) a.drho[i] = orr;
  }
}
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const BwdArgs a) { wgrad_finish_body(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void bwd_finish_pair_kernel(const BwdArgs a) {   // blocks [0, fin_nw): wgrad's finishing pass, the rest dgrad's
  if ((int)blockIdx.x < a.fin_nw) wgrad_finish_body(a, blockIdx.x, a.fin_nw);
  else dgrad_finish_body(a, (int)blockIdx.x - a.fin_nw, (int)gridDim.x - a.fin_nw);
}

// dKL/dmu, dKL/drho of the normal-prior KL (base_variational_layer.py:70-72), scaled by the upstream gradient g[0] / n
__global__ __launch_bounds__(256) void kl_normal_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ rho, const float* __restrict__ pmu,
                                                            const float* __restrict__ psig, const float* __restrict__ gup, long long n, int laplace,
                                                            float* __restrict__ dmu, float* __restrict__ drho) {
  const float gs = gup[0] / (float)n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    kl_elem_grad(mu[i], rho[i], pmu, psig, i, laplace, gs, dmu[i], drho[i]);
  }
}

// Workgroups per output tile: first the samples (each group keeps whole samples), then chunks of the reduction axis
// M = B*Ho*Wo (a training step has ONE sample: layer1's 9 output tiles would otherwise be 9 workgroups on 256 CUs).
// Chunks are multiples of the LDS stage and at least 8 stages long.
// All tensors of a model in one launch (get_kl_loss under autograd): block b works on 1024 elements of segment seg(b).
struct KlBwdSegs {
  const float *mu[BT_KL_MAX_SEGMENTS], *rho[BT_KL_MAX_SEGMENTS], *pmu[BT_KL_MAX_SEGMENTS], *psig[BT_KL_MAX_SEGMENTS];
  float *dmu[BT_KL_MAX_SEGMENTS], *drho[BT_KL_MAX_SEGMENTS];
  long long n[BT_KL_MAX_SEGMENTS];
  int boff[BT_KL_MAX_SEGMENTS + 1];  // first block of every segment
  int nseg, laplace;
};
__global__ __launch_bounds__(256) void kl_normal_bwd_segs_kernel(const KlBwdSegs a, const float* __restrict__ gup) {
  int seg = 0;
  while (seg + 1 < a.nseg && (int)blockIdx.x >= a.boff[seg + 1]) ++seg;   // uniform
  const long long n = a.n[seg];
  const float gs = gup[0] / (float)n;
  const float *mu = a.mu[seg], *rho = a.rho[seg], *pmu = a.pmu[seg], *psig = a.psig[seg];
  float *dmu = a.dmu[seg], *drho = a.drho[seg];
  const long long base = (long long)((int)blockIdx.x - a.boff[seg]) * 1024;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + threadIdx.x + 256 * k;
    if (i < n) {
      const float m = mu[i], r = rho[i], sq = softplus(r);
      const float sg = __builtin_amdgcn_rcpf(__fadd_rn(1.0f, __builtin_amdgcn_exp2f(__fmul_rn(-1.4426950408889634f, r))));
      float gm, gq;
      if (a.laplace) {
        const float z = __fmul_rn(m, __builtin_amdgcn_rcpf(__fmul_rn(sq, 1.4142135623730951f)));
        gm = erff(z);
        gq = __fsub_rn(__fmul_rn(0.7978845608028654f, __builtin_amdgcn_exp2f(__fmul_rn(-1.4426950408889634f, __fmul_rn(z, z)))), __builtin_amdgcn_rcpf(sq));
      } else {
        const float ip = __builtin_amdgcn_rcpf(__fmul_rn(psig[i], psig[i]));
        gm = __fmul_rn(__fsub_rn(m, pmu[i]), ip);
        gq = __fsub_rn(__fmul_rn(sq, ip), __builtin_amdgcn_rcpf(sq));
      }
      dmu[i] = __fmul_rn(gm, gs);
      drho[i] = __fmul_rn(__fmul_rn(gq, sg), gs);
    }
  }
}

static int wgrad_groups(const bt_conv2d_geom& g, int S, int* sgroups = nullptr, int* mchunk = nullptr) {
  const int Cig = g.Ci / g.groups, Cog = g.Co / g.groups, Cig4 = (Cig + 3) & ~3, T = g.kh * g.kw;
  const long long tiles = (long long)((T * Cig4 + 63) / 64) * ((Cog + 63) / 64) * g.groups;
  static const int w_target = [] { const char* e = getenv("BT_WGRAD_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();   // measurement knob
  // One workgroup per CU: the pass runs beside dgrad in one launch (bwd_pair_kernel) and every group costs a partial that the finishing
  // pass reads back (ResNet18 step, env sweep: 512 -> 3.14 ms, 256 -> 2.97, 192 -> 2.98, 128 -> 3.24, 1024 -> 3.22).
  long long gr = (w_target + tiles - 1) / tiles;
  if (gr < 1) gr = 1;
  const long long gs = gr > S ? S : gr;
  const int Ho = (g.H + 2 * g.ph - g.dh * (g.kh - 1) - 1) / g.sh + 1, Wo = (g.W + 2 * g.pw - g.dw * (g.kw - 1) - 1) / g.sw + 1;
  const long long M = (long long)g.B * Ho * Wo;
  long long gm = (gr + gs - 1) / gs;
  const long long max_gm = (M + 8 * kBK - 1) / (8 * kBK);
  if (gm > max_gm) gm = max_gm;
  if (gm < 1) gm = 1;
  long long chunk = (M + gm - 1) / gm;
  chunk = (chunk + kBK - 1) / kBK * kBK;
  gm = (M + chunk - 1) / chunk;
  if (sgroups) *sgroups = (int)gs;
  if (mchunk) *mchunk = (int)chunk;
  return (int)(gs * gm);
}

// dgrad: pieces of the output-channel reduction, so that a launch offers ~512 workgroups (layer4 of a CIFAR ResNet at one
// sample: 16 tiles); pieces are multiples of the LDS stage and at least 32 channels.
static int dgrad_chunks(const bt_conv2d_geom& g, int S, int* dchunk = nullptr) {
  const int Cig = g.Ci / g.groups, Cog = g.Co / g.groups;
  const long long M = (long long)g.B * g.H * g.W;
  const long long tiles = ((M + 63) / 64) * ((Cig + 63) / 64) * S * g.groups;
  static const int d_target = [] { const char* e = getenv("BT_DGRAD_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();   // measurement knob
  long long c = (d_target + tiles - 1) / tiles;
  const long long max_c = (Cog + 31) / 32;
  if (c > max_c) c = max_c;
  if (c < 1) c = 1;
  long long per = (Cog + c - 1) / c;
  per = (per + kBK - 1) / kBK * kBK;
  c = (Cog + per - 1) / per;
  if (dchunk) *dchunk = (int)per;
  return (int)c;
}

}  // namespace bt

extern "C" size_t bt_conv2d_bwd_workspace(const bt_conv2d_geom* g, int32_t S) {
  if (!g || S <= 0 || g->groups <= 0) return 0;
  const int Cig = g->Ci / g->groups, Cig4 = (Cig + 3) & ~3, T = g->kh * g->kw;
  const size_t wg = (size_t)2 * bt::wgrad_groups(*g, S) * g->Co * T * Cig4 * sizeof(float);
  const int dc = bt::dgrad_chunks(*g, S);
  const size_t dg = dc > 1 ? (size_t)dc * S * g->B * g->Ci * g->H * g->W * sizeof(float) : 0;
  return wg + dg;   // (wgrad's partials, then dgrad's: the two passes run side by side in one launch when both are asked for)
}

static int conv2d_bwd_impl(const bt_conv2d_geom* g, int32_t S, int32_t flipout, const float* x, int64_t x_sample_stride, const float* grad_out,
                           const bt_params* p, const bt_draws* d, const float* grad_kl, float* dx, float* dmu_w, float* drho_w, void* workspace,
                           size_t workspace_bytes, bt_stream_t stream) {
  using namespace bt;
  if (!g || !x || !grad_out || !p || !d) return set_error(BT_ERR_BAD_ARG, "bt_conv2d_bwd: null argument");
  if (grad_kl) {
    const bool lap = p->prior_kind == BT_PRIOR_LAPLACE;
    if (!dmu_w) return set_error(BT_ERR_BAD_ARG, "bt_conv2d_bwd_kl: grad_kl needs dmu_w / drho_w");
    if (!p->mu_w || (!lap && (!p->prior_mu_w || !p->prior_sigma_w))) return set_error(BT_ERR_BAD_ARG, "bt_conv2d_bwd_kl: needs mu_w and the weight priors");
  }
  if (!p->mu_packed || !p->sigma_packed || !p->rho_w) return set_error(BT_ERR_BAD_ARG, "bt_conv2d_bwd: needs rho_w and the packed parameters (bt_pack_params)");
  if (S <= 0 || g->B <= 0 || g->Ci <= 0 || g->Co <= 0 || g->groups <= 0 || g->Ci % g->groups || g->Co % g->groups) return set_error(BT_ERR_BAD_ARG, "bt_conv2d_bwd: bad geometry");
  if ((dmu_w == nullptr) != (drho_w == nullptr)) return set_error(BT_ERR_BAD_ARG, "bt_conv2d_bwd: dmu_w and drho_w go together");
  const bool inj = d->eps_w != nullptr;
  if (flipout && inj != (d->sign_in != nullptr && d->sign_out != nullptr)) return set_error(BT_ERR_BAD_ARG, "bt_conv2d_bwd: inject all draws or none");
  BwdArgs a = {};
  a.x = x, a.g = grad_out, a.mu_pk = p->mu_packed, a.sig_pk = p->sigma_packed, a.rho_w = p->rho_w;
  a.eps_w = d->eps_w, a.sign_in = flipout ? d->sign_in : nullptr, a.sign_out = flipout ? d->sign_out : nullptr;
  a.dx = dx, a.dmu = dmu_w, a.drho = drho_w, a.part = (float*)workspace;
  if (grad_kl) a.gkl = grad_kl, a.mu_w = p->mu_w, a.pmu_w = p->prior_mu_w, a.psig_w = p->prior_sigma_w, a.kl_laplace = p->prior_kind == BT_PRIOR_LAPLACE ? 1 : 0;
  a.B = g->B, a.Ci = g->Ci, a.H = g->H, a.W = g->W, a.Co = g->Co, a.KH = g->kh, a.KW = g->kw;
  a.SH = g->sh, a.SW = g->sw, a.PH = g->ph, a.PW = g->pw, a.DH = g->dh, a.DW = g->dw, a.G = g->groups;
  a.Ho = (g->H + 2 * g->ph - g->dh * (g->kh - 1) - 1) / g->sh + 1;
  a.Wo = (g->W + 2 * g->pw - g->dw * (g->kw - 1) - 1) / g->sw + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return set_error(BT_ERR_BAD_ARG, "bt_conv2d_bwd: empty output");
  a.Cig = g->Ci / g->groups, a.Cog = g->Co / g->groups, a.Cig4 = (a.Cig + 3) & ~3, a.T = g->kh * g->kw, a.S = S, a.flip = flipout ? 1 : 0;
  a.x_sample_stride = x_sample_stride;
  a.x_elems = (long long)g->B * g->Ci * g->H * g->W;
  a.out_elems = (long long)g->B * g->Co * a.Ho * a.Wo;
  a.w_elems = (long long)g->Co * a.Cig * a.T;
  if (a.x_elems >= (1ll << 31) || a.out_elems >= (1ll << 31) || a.w_elems * 4 >= (1ll << 31)) return set_error(BT_ERR_UNSUPPORTED, "bt_conv2d_bwd: tensor too large for the 32-bit sign / draw indices");
  a.seed_lo = (uint32_t)d->rng.seed, a.seed_hi = (uint32_t)(d->rng.seed >> 32);
  a.call = d->rng.call, a.call_base = d->rng.call_base_dev, a.layer_id = d->rng.layer_id, a.sample0 = d->rng.sample0;
  hipStream_t st = (hipStream_t)stream;
  const size_t need_ws = bt_conv2d_bwd_workspace(g, S);
  dim3 dgrid(1, 1, 1), wgrid(1, 1, 1);
  long long n_dfin = 0, n_wfin = 0;
  if (dx) {
    const long long M = (long long)g->B * g->H * g->W;
    a.dchunks = dgrad_chunks(*g, S, &a.dchunk);
    if (a.dchunks > 1 && (!workspace || workspace_bytes < need_ws))
      return set_error(BT_ERR_WORKSPACE, "bt_conv2d_bwd: workspace smaller than bt_conv2d_bwd_workspace()");
    dgrid = dim3((unsigned)((M + 63) / 64), (unsigned)((a.Cig + 63) / 64), (unsigned)(S * g->groups * a.dchunks));
    if (a.dchunks > 1) n_dfin = ((long long)S * a.x_elems + 255) / 256 > 4096 ? 4096 : ((long long)S * a.x_elems + 255) / 256;
  }
  if (dmu_w) {
    a.groups = wgrad_groups(*g, S, &a.sgroups, &a.mchunk);
    if (!workspace || workspace_bytes < need_ws) return set_error(BT_ERR_WORKSPACE, "bt_conv2d_bwd: workspace smaller than bt_conv2d_bwd_workspace()");
    wgrid = dim3((unsigned)((a.T * a.Cig4 + 63) / 64), (unsigned)((a.Cog + 63) / 64), (unsigned)(a.groups * g->groups));
    n_wfin = (a.w_elems + 255) / 256 > 4096 ? 4096 : (a.w_elems + 255) / 256;
  }
  // dgrad's partials live behind wgrad's
  a.part_d = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (size_t)2 * wgrad_groups(*g, S) * g->Co * a.T * a.Cig4 * sizeof(float));
  const long long nwb = (long long)wgrid.x * wgrid.y * wgrid.z, ndb = (long long)dgrid.x * dgrid.y * dgrid.z;
  static const bool no_pair = [] { const char* e = getenv("BT_NO_BWD_PAIR"); return e && *e && *e != '0'; }();   // measurement knob: the two passes as launches of their own
  if (!no_pair && dx && dmu_w && nwb + ndb < 0x7FFFFFFFll) {   // both passes: ONE launch, and one for the two finishing passes
    a.pair_wx = (int)wgrid.x, a.pair_wy = (int)wgrid.y, a.pair_nw = (int)nwb, a.pair_dx = (int)dgrid.x, a.pair_dy = (int)dgrid.y;
    if (flipout) hipLaunchKernelGGL(bwd_pair_kernel<true>, dim3((unsigned)(nwb + ndb)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(bwd_pair_kernel<false>, dim3((unsigned)(nwb + ndb)), dim3(256), 0, st, a);
    if (int rc = check_launch("bt_conv2d_bwd (wgrad + dgrad)")) return rc;
    a.fin_nw = (int)n_wfin;
    hipLaunchKernelGGL(bwd_finish_pair_kernel, dim3((unsigned)(n_wfin + n_dfin)), dim3(256), 0, st, a);
    return check_launch("bt_conv2d_bwd (finish)");
  }
  if (dx) {
    if (flipout) hipLaunchKernelGGL(dgrad_kernel<true>, dgrid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dgrad_kernel<false>, dgrid, dim3(256), 0, st, a);
    if (int rc = check_launch("bt_conv2d_bwd (dgrad)")) return rc;
    if (a.dchunks > 1) {
      hipLaunchKernelGGL(dgrad_finish_kernel, dim3((unsigned)n_dfin), dim3(256), 0, st, a);
      if (int rc = check_launch("bt_conv2d_bwd (dgrad finish)")) return rc;
    }
  }
  if (dmu_w) {
    if (flipout) hipLaunchKernelGGL(wgrad_kernel<true>, wgrid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wgrad_kernel<false>, wgrid, dim3(256), 0, st, a);
    if (int rc = check_launch("bt_conv2d_bwd (wgrad)")) return rc;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)n_wfin), dim3(256), 0, st, a);
    if (int rc = check_launch("bt_conv2d_bwd (finish)")) return rc;
  }
  return BT_OK;
}

extern "C" int bt_conv2d_bwd(const bt_conv2d_geom* g, int32_t S, int32_t flipout, const float* x, int64_t x_sample_stride, const float* grad_out,
                             const bt_params* p, const bt_draws* d, float* dx, float* dmu_w, float* drho_w, void* workspace, size_t workspace_bytes,
                             bt_stream_t stream) {
  return conv2d_bwd_impl(g, S, flipout, x, x_sample_stride, grad_out, p, d, nullptr, dx, dmu_w, drho_w, workspace, workspace_bytes, stream);
}

extern "C" int bt_conv2d_bwd_kl(const bt_conv2d_geom* g, int32_t S, int32_t flipout, const float* x, int64_t x_sample_stride, const float* grad_out,
                                const bt_params* p, const bt_draws* d, const float* grad_kl, float* dx, float* dmu_w, float* drho_w, void* workspace,
                                size_t workspace_bytes, bt_stream_t stream) {
  return conv2d_bwd_impl(g, S, flipout, x, x_sample_stride, grad_out, p, d, grad_kl, dx, dmu_w, drho_w, workspace, workspace_bytes, stream);
}

extern "C" int bt_kl_normal_bwd(const float* mu, const float* rho, const float* prior_mu, const float* prior_sigma, const float* grad_kl, int64_t numel,
                                uint32_t flags, float* dmu, float* drho, bt_stream_t stream) {
  using namespace bt;
  if (!mu || !rho || !grad_kl || !dmu || !drho || numel <= 0) return set_error(BT_ERR_BAD_ARG, "bt_kl_normal_bwd: null argument");
  const int lap = (flags & BT_KL_PRIOR_LAPLACE) ? 1 : 0;
  if (!lap && (!prior_mu || !prior_sigma)) return set_error(BT_ERR_BAD_ARG, "bt_kl_normal_bwd: priors are required for the normal prior");
  const long long nb = (numel + 255) / 256;
  hipLaunchKernelGGL(kl_normal_bwd_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, (hipStream_t)stream, mu, rho, prior_mu, prior_sigma, grad_kl,
                     (long long)numel, lap, dmu, drho);
  return check_launch("bt_kl_normal_bwd");
}

extern "C" int bt_kl_normal_bwd_segs(int32_t n_segments, const float* const* mu, const float* const* rho, const float* const* prior_mu,
                                     const float* const* prior_sigma, const int64_t* numel, const float* grad_kl, uint32_t flags,
                                     float* const* dmu, float* const* drho, bt_stream_t stream) {
  using namespace bt;
  if (n_segments <= 0 || n_segments > BT_KL_MAX_SEGMENTS || !mu || !rho || !numel || !grad_kl || !dmu || !drho)
    return set_error(BT_ERR_BAD_ARG, "bt_kl_normal_bwd_segs: bad argument");
  const int lap = (flags & BT_KL_PRIOR_LAPLACE) ? 1 : 0;
  if (!lap && (!prior_mu || !prior_sigma)) return set_error(BT_ERR_BAD_ARG, "bt_kl_normal_bwd_segs: priors are required for the normal prior");
  KlBwdSegs a = {};
  long long blocks = 0;
  for (int i = 0; i < n_segments; ++i) {
    if (!mu[i] || !rho[i] || !dmu[i] || !drho[i] || numel[i] <= 0 || (!lap && (!prior_mu[i] || !prior_sigma[i])))
      return set_error(BT_ERR_BAD_ARG, "bt_kl_normal_bwd_segs: null tensor or empty segment");
    a.mu[i] = mu[i], a.rho[i] = rho[i], a.pmu[i] = lap ? nullptr : prior_mu[i], a.psig[i] = lap ? nullptr : prior_sigma[i];
    a.dmu[i] = dmu[i], a.drho[i] = drho[i], a.n[i] = numel[i];
    a.boff[i] = (int)blocks;
    blocks += (numel[i] + 1023) / 1024;
    if (blocks > 0x7FFFFFFFll) return set_error(BT_ERR_UNSUPPORTED, "bt_kl_normal_bwd_segs: too many elements for one launch");
  }
  a.boff[n_segments] = (int)blocks;
  a.nseg = n_segments, a.laplace = lap;
  hipLaunchKernelGGL(kl_normal_bwd_segs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, grad_kl);
  return check_launch("bt_kl_normal_bwd_segs");
}
