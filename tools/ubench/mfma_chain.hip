// Micro-experiment: does a chain of dependent v_mfma_f32_32x32x16_bf16 (same accumulator back to back) issue at the rate of
// independent ones? One wave per SIMD (256 threads), one block per CU. NCH accumulators used round-robin; 24 MFMAs per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int NCH>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
  f32x16 acc[NCH];
  for (int i = 0; i < NCH; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (short)(threadIdx.x + j); bb[j] = (short)(threadIdx.x * 3 + j); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 24 / NCH; ++r)
#pragma unroll
      for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < NCH; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NCH> void run(float* d, unsigned long long* c, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NCH><<<256, 256>>>(d, iters, c); hipDeviceSynchronize();
  hipEventRecord(e0); k<NCH><<<256, 256>>>(d, iters, c); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("chains %d: %.3f ms  %.1f s_memtime ticks per MFMA  (ticks/us %.0f)\n", NCH, ms, (double)h / (24.0 * iters), (double)h / (ms * 1e3));
}
int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4); unsigned long long* c; hipMalloc(&c, 8);
  const int iters = 20000;
  run<1>(d, c, iters); run<2>(d, c, iters); run<3>(d, c, iters); run<4>(d, c, iters); run<8>(d, c, iters);
  return 0;
}
