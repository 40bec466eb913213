// Micro-experiment: do fp32-input MFMA and VALU work co-execute on one SIMD of gfx950?
// Kernel modes (block = 512 threads = 8 waves, 2 per SIMD; grid = 256 blocks, one per CU):
//   0: waves 0-3 run an MFMA-f32 loop, waves 4-7 idle      1: waves 4-7 run a VALU loop, waves 0-3 idle
//   2: both                                                 3/4/5: same with bf16 MFMA
//   6: VALU = v_mad_u64_u32 chain only (throughput)         7: VALU = v_fma_f32 only
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int VK>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  constexpr bool BF = (MODE >= 3 && MODE <= 5);
  const bool do_mfma = (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 5);
  const bool do_valu = (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7);
  if (wave < 4) {
    if (!do_mfma) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f - a;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (short)(threadIdx.x + j); bb[j] = (short)(threadIdx.x * 3 + j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!do_valu) return;
    if (VK == 2) {
      unsigned long long x = threadIdx.x * 0x9E3779B97F4A7C15ull + 1;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { x = (unsigned long long)(unsigned)x * 0xD2511F53u + (x >> 32); }
      }
      out[blockIdx.x * 512 + threadIdx.x] = (float)x;
    } else if (VK == 1) {
      unsigned v[8];
      for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
      }
      unsigned s = 0; for (int i = 0; i < 8; ++i) s += v[i];
      out[blockIdx.x * 512 + threadIdx.x] = (float)s;
    } else if (VK == 3) {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      }
      float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(0.5f));
      }
      float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    }
  }
}

template <int MODE, int VK> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  const int it = 20000;
  printf("per iteration: MFMA wave 4 MFMA (f32 4x64 cyc, bf16 4x32 cyc); VALU wave 16 instr\n");
  printf("mfma-f32 only %.3f   mfma-bf16 only %.3f\n", run<0,0>(d, it), run<3,0>(d, it));
  const char* nm[4] = {"v_fma_f32", "v_xor_b32", "v_mad_u64_u32", "v_exp_f32"};
  float a0 = run<1,0>(d, it), a1 = run<1,1>(d, it), a2 = run<1,2>(d, it), a3 = run<1,3>(d, it);
  float f0 = run<2,0>(d, it), f1 = run<2,1>(d, it), f2 = run<2,2>(d, it), f3 = run<2,3>(d, it);
  float b0 = run<5,0>(d, it), b1 = run<5,1>(d, it), b2 = run<5,2>(d, it), b3 = run<5,3>(d, it);
  float al[4] = {a0,a1,a2,a3}, fl[4] = {f0,f1,f2,f3}, bl[4] = {b0,b1,b2,b3};
  for (int i = 0; i < 4; ++i) printf("%-14s alone %.3f   + mfma-f32 %.3f   + mfma-bf16 %.3f\n", nm[i], al[i], fl[i], bl[i]);
  return 0;
}
