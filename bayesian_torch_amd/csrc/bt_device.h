// Device-side building blocks shared by every kernel of libbtorch_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bt {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------- Philox4x32-10
// Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11). Counter-based: the draw
// for (tensor element, sample, layer, call) is a pure function of its coordinates, so any block
// can regenerate any weight tile without state and MC samples can be sharded arbitrarily.
__host__ __device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1;
  c[3] = (uint32_t)p0;
  c[0] = n0;
  c[2] = n2;
}

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// Coordinates of one draw stream. tensor: 0 eps_w, 1 eps_b, 2 sign_in, 3 sign_out.
struct RngKey {
  uint32_t seed_lo, seed_hi, call, layer_tensor;  // layer_tensor = layer_id | tensor << 28
};

__host__ __device__ __forceinline__ uint32_t layer_tensor_word(uint32_t layer_id, uint32_t tensor) {
  return (layer_id & 0x0FFFFFFFu) | (tensor << 28);
}

// 4 raw words for elements [4q, 4q+3] of (sample, layer, tensor, call).
__device__ __forceinline__ void philox_block(const RngKey& k, uint32_t sample, uint32_t q, uint32_t (&r)[4]) {
  r[0] = q;
  r[1] = sample;
  r[2] = k.layer_tensor;
  r[3] = k.call;
  philox4x32_10(r, k.seed_lo, k.seed_hi);
}

// Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin/v_cos (input in
// revolutions). No a*b+c shapes below, so -ffp-contract cannot make two call sites differ.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u = __fadd_rn(__fmul_rn((float)(a >> 8), 5.9604644775390625e-8f), 2.98023223876953125e-8f);  // (0, 1]
  const float t = __fmul_rn((float)b, 2.3283064365386963e-10f);                                             // [0, 1] revolutions
  const float r = __builtin_amdgcn_sqrtf(__fmul_rn(-1.3862943611198906f, __builtin_amdgcn_logf(u)));      // sqrt(-2 ln u)
  z0 = __fmul_rn(r, __builtin_amdgcn_cosf(t));
  z1 = __fmul_rn(r, __builtin_amdgcn_sinf(t));
}

__device__ __forceinline__ void philox_normal4(const RngKey& k, uint32_t sample, uint32_t q, float (&z)[4]) {
  uint32_t r[4];
  philox_block(k, sample, q, r);
  box_muller(r[0], r[1], z[0], z[1]);
  box_muller(r[2], r[3], z[2], z[3]);
}

// Flipout signs: activation-sized streams, consumed once per im2col tap, so a full Philox per
// element would cost more than the contraction it decorates. The per-(seed, call, layer, tensor,
// sample) key is derived once per block with Philox; each element then takes one round of a
// low-bias 32-bit mixer (Wellons' "lowbias32") of (index ^ key) and uses its top bit.
__device__ __forceinline__ uint32_t sign_stream_key(const RngKey& k, uint32_t sample) {
  uint32_t r[4];
  philox_block(k, sample, 0xFFFFFFFFu, r);
  return r[0] ^ r[2];
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

__device__ __forceinline__ float hash_sign(uint32_t key, uint32_t idx) {
  // two dependent mixes: a single one leaves visible structure between idx and idx ^ key patterns
  const uint32_t h = mix32(mix32(idx ^ key) + key);
  return __uint_as_float(0x3F800000u | (h & 0x80000000u));  // +1.0f or -1.0f
}

// ---------------------------------------------------------------------------- softplus / KL
// sigma = log1p(exp(rho)) exactly as the reference writes it (no threshold; overflows to inf with it).
// log1p through the u = 1 + e correction so tiny sigma keeps full relative accuracy.
__device__ __forceinline__ float softplus(float rho) {
  const float e = __builtin_amdgcn_exp2f(__fmul_rn(rho, 1.4426950408889634f));
  const float u = __fadd_rn(1.0f, e);
  const float d = __fsub_rn(u, 1.0f);
  const float lu = __fmul_rn(__builtin_amdgcn_logf(u), 0.6931471805599453f);
  float s = __fmul_rn(lu, __fmul_rn(e, __builtin_amdgcn_rcpf(d)));
  s = (d == 0.0f) ? e : s;
  s = (e > 16777216.0f) ? lu : s;
  return s;
}

__device__ __forceinline__ float ln_fast(float x) { return __fmul_rn(__builtin_amdgcn_logf(x), 0.6931471805599453f); }

// One element of kl_div's 'normal' branch (base_variational_layer.py:70-71), same op order.
__device__ __forceinline__ float kl_term(float mu_q, float sigma_q, float mu_p, float sigma_p) {
  const float dm = __fsub_rn(mu_q, mu_p);
  const float num = __fadd_rn(__fmul_rn(sigma_q, sigma_q), __fmul_rn(dm, dm));
  const float den = __fmul_rn(2.0f, __fmul_rn(sigma_p, sigma_p));
  const float q = __fmul_rn(num, __builtin_amdgcn_rcpf(den));
  return __fsub_rn(__fadd_rn(__fsub_rn(ln_fast(sigma_p), ln_fast(sigma_q)), q), 0.5f);
}

// One element of kl_div's 'laplace' branch (base_variational_layer.py:74-97): KL(N(mu_q, sigma_q^2) || Laplace(0, 1)) --
// the reference hard-codes the prior's location 0 and scale 1 there, whatever prior tensors it is handed --
//   log 2 - 0.5 log(2 pi sigma^2) - 0.5 + E|w|,   E|w| = sigma sqrt(2/pi) exp(-mu^2 / (2 sigma^2)) + mu (1 - 2 Phi(-mu/sigma)),
// with 1 - 2 Phi(-z) = erf(z / sqrt 2).
__device__ __forceinline__ float kl_term_laplace(float mu_q, float sigma_q) {
  const float s2 = __fmul_rn(sigma_q, sigma_q);
  const float z2 = __fmul_rn(__fmul_rn(mu_q, mu_q), __builtin_amdgcn_rcpf(__fmul_rn(2.0f, s2)));
  const float g = __builtin_amdgcn_exp2f(__fmul_rn(-1.4426950408889634f, z2));
  const float z = __fmul_rn(mu_q, __builtin_amdgcn_rcpf(__fmul_rn(sigma_q, 1.4142135623730951f)));
  const float e_abs = __fadd_rn(__fmul_rn(__fmul_rn(sigma_q, 0.7978845608028654f), g), __fmul_rn(mu_q, erff(z)));
  const float head = __fsub_rn(__fsub_rn(0.6931471805599453f, __fmul_rn(0.5f, ln_fast(__fmul_rn(6.283185307179586f, s2)))), 0.5f);
  return __fadd_rn(head, e_abs);
}

// ---------------------------------------------------------------------------- reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over a 256-thread block; result valid in thread 0. `scratch` = 4 doubles of LDS.
__device__ __forceinline__ double block_sum_256(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  return t;
}

// Deterministic cross-block finish: every contributing block publishes one double, takes a
// ticket, and the last arriver sums the slots in index order. Protocol per
// cdna_hip_programming.md Guideline 16 (agent-scope release before the ticket, agent-scope
// acquire in the last arriver, explicit vmcnt drains: ROCm 7.2 can drop the fence's own wait).
// Call from thread 0 of the block only. Returns true in the last arriver (after the acquire).
__device__ __forceinline__ bool publish_and_ticket(double* slots, unsigned* counter, int slot, double v, unsigned expected) {
  slots[slot] = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t != expected - 1u) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return true;
}

// The same hand-off without any fence: the partial goes out as an agent-scope (sc1, write-through) 8-byte store, the storing lane
// drains it (vmcnt(0)) before its agent-scope ticket add, and the last arriver -- told by the value its own add returned -- reads
// the slots with agent-scope (sc1) loads, which bypass its L1 (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 table:
// one unsharded counter, all handed-off bytes stored and loaded sc1). No L2 write-back, so the cost does not depend on how much
// output other workgroups of the XCD have dirtied by then: the partial can be published at any point of the kernel.
__device__ __forceinline__ bool publish_and_ticket_wt(double* slots, unsigned* counter, int slot, double v, unsigned expected) {
  __hip_atomic_store(&slots[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return t == expected - 1u;
}

// Workspace layout (BT_WORKSPACE_BYTES = 64 KiB, zero-filled by the caller once):
//   [0, 8)        ticket counter (uint32) + pad
//   [64, 64+8*N)  double slots, N <= kMaxSlots
constexpr int kMaxSlots = 8000;
__host__ __device__ inline unsigned* ws_counter(void* ws) { return reinterpret_cast<unsigned*>(ws); }
__host__ __device__ inline double* ws_slots(void* ws) { return reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + 64); }

}  // namespace bt
