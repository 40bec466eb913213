/*
 * CPU ORACLE (plain C) -- test infrastructure, NOT product code.
 *
 * An independent restatement of the reference's stochastic variational-layer forward + KL in plain C, with no
 * dependency on ATen: the element-wise steps are done in fp32 exactly in the reference's order (sigma = log1p(exp(rho));
 * tmp = sigma*eps; w = mu + tmp; KL term), the contraction and the KL mean accumulate in fp64 (order-independent at
 * fp32 resolution, so it sits between MKL-DNN's and the MFMA's summation orders). Checked against the golden vectors
 * produced by running the reference (tests/test_oracle_golden.py) -- parity PINNED -- and used by the GPU tests as a second
 * checker next to oracle/bt_oracle.py.  Built by oracle/Makefile into oracle/libbt_oracle_c.so (ctypes).
 *
 * Reference lines followed (under /root/reference/bayesian_torch/):
 *   sampling        layers/variational_layers/linear_variational.py:163-166, conv_variational.py:366-369
 *   KL normal       layers/base_variational_layer.py:70-72
 *   KL laplace      layers/base_variational_layer.py:74-97 (prior location 0 and scale 1 hard-coded there)
 *   conv / linear   F.conv2d / F.linear call sites: conv_variational.py:384, linear_variational.py:181
 *   flipout         layers/flipout_layers/conv_flipout.py:376-417, linear_flipout.py:149-174
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static float softplus_f(float rho) { return log1pf(expf(rho)); }

/* mean_i( log sp - log sq + (sq^2 + (mq - mp)^2) / (2 sp^2) - 0.5 ), sq = softplus(rho) */
double bto_kl_normal(const float *mu, const float *rho, const float *pmu, const float *psig, int64_t n) {
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const float sq = softplus_f(rho[i]);
    const float d = mu[i] - pmu[i];
    const float t = logf(psig[i]) - logf(sq) + (sq * sq + d * d) / (2.0f * (psig[i] * psig[i])) - 0.5f;
    acc += (double)t;
  }
  return acc / (double)n;
}

/* mean_i( log 2 - 0.5 log(2 pi sq^2) - 0.5 + E|w| ),  E|w| = sq sqrt(2/pi) exp(-mu^2 / (2 sq^2)) + mu (1 - 2 Phi(-mu/sq)),
 * Phi(-z) = erfc(z / sqrt 2) / 2  =>  1 - 2 Phi(-z) = erf(z / sqrt 2). Per element in fp64 (an independent check of the
 * fp32 op chain), mean in fp64. */
double bto_kl_laplace(const float *mu, const float *rho, int64_t n) {
  const double two_over_pi = 0.63661977236758134308, two_pi = 6.28318530717958647692;
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double sq = (double)softplus_f(rho[i]), m = (double)mu[i];
    const double e_abs = sq * sqrt(two_over_pi) * exp(-m * m / (2.0 * sq * sq)) + m * erf(m / (sq * 1.41421356237309504880));
    acc += log(2.0) - 0.5 * log(two_pi * sq * sq) - 0.5 + e_abs;
  }
  return acc / (double)n;
}

/* w = mu + softplus(rho) * eps   (two fp32 roundings); delta != NULL also receives softplus(rho) * eps */
void bto_sample(const float *mu, const float *rho, const float *eps, int64_t n, float *w, float *delta) {
  for (int64_t i = 0; i < n; ++i) {
    const float tmp = softplus_f(rho[i]) * eps[i];
    if (delta) delta[i] = tmp;
    if (w) w[i] = mu[i] + tmp;
  }
}

/* out[b][co][ho][wo] = bias[co] + sum x[b][g*Cig+ci][ho*sh-ph+kh*dh][wo*sw-pw+kw*dw] * w[co][ci][kh][kw]  (NCHW, fp64 accumulate) */
void bto_conv2d(const float *x, const float *w, const float *bias, float *out, int B, int Ci, int H, int W, int Co, int kh, int kw,
                int sh, int sw, int ph, int pw, int dh, int dw, int groups) {
  const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
  const int Cig = Ci / groups, Cog = Co / groups;
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Co; ++co) {
      const int g = co / Cog;
      for (int ho = 0; ho < Ho; ++ho)
        for (int wo = 0; wo < Wo; ++wo) {
          double acc = 0.0;
          for (int ci = 0; ci < Cig; ++ci)
            for (int a = 0; a < kh; ++a) {
              const int hi = ho * sh - ph + a * dh;
              if (hi < 0 || hi >= H) continue;
              for (int c = 0; c < kw; ++c) {
                const int wi = wo * sw - pw + c * dw;
                if (wi < 0 || wi >= W) continue;
                acc += (double)x[(((size_t)b * Ci + g * Cig + ci) * H + hi) * W + wi] * (double)w[(((size_t)co * Cig + ci) * kh + a) * kw + c];
              }
            }
          out[(((size_t)b * Co + co) * Ho + ho) * Wo + wo] = (float)(acc + (bias ? (double)bias[co] : 0.0));
        }
    }
}

/* out[b][o] = bias[o] + sum_k x[b][k] * w[o][k] */
void bto_linear(const float *x, const float *w, const float *bias, float *out, int B, int In, int Out) {
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < Out; ++o) {
      double acc = 0.0;
      for (int k = 0; k < In; ++k) acc += (double)x[(size_t)b * In + k] * (double)w[(size_t)o * In + k];
      out[(size_t)b * Out + o] = (float)(acc + (bias ? (double)bias[o] : 0.0));
    }
}

/* flipout combine: out = mean + pert * s_out ;  xs = x * s_in */
void bto_mul(const float *a, const float *b, float *o, int64_t n) { for (int64_t i = 0; i < n; ++i) o[i] = a[i] * b[i]; }
void bto_fma_sign(const float *mean, const float *pert, const float *s_out, float *o, int64_t n) {
  for (int64_t i = 0; i < n; ++i) o[i] = mean[i] + pert[i] * s_out[i];
}
