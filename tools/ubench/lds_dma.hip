// Semantics probe for buffer_load_dwordx4 ... lds on gfx950: lane -> LDS placement, what an out-of-range lane writes,
// whether exec-masked lanes write, and the scalar offset operand.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ubench/lds_dma tools/ubench/lds_dma.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const float* x, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 2048; i += 256) smem[i] = 7.0f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
  const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) * 64);
  // odd lanes out of range; lanes >= 200 masked off; source chunk = lane (reversed inside each wave to show per-lane addresses)
  const int lane = threadIdx.x & 63;
  const uint32_t off = (threadIdx.x & 1) ? 0x80000000u : (uint32_t)(wbase + 63 - lane) * 16u;
  if (threadIdx.x < 200)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 4 * wbase), 16, (int)off, 4096, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) out[i] = smem[i];
}
int main() {
  const int n = 4096;
  float *x, *out, hx[n], ho[2048];
  for (int i = 0; i < n; ++i) hx[i] = (float)i;
  hipMalloc(&x, n * 4), hipMalloc(&out, 2048 * 4);
  hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 2048 * 4, 0, x, out, n);
  hipMemcpy(ho, out, 2048 * 4, hipMemcpyDeviceToHost);
  // expectations: even thread t < 200: smem[4t..4t+3] = x[1024 + 4*(wbase + 63 - lane) + 0..3]; odd t < 200: 0 (or 7 if the write is skipped); t >= 200: 7
  int bad = 0, oob_zero = 0, oob_skip = 0;
  for (int t = 0; t < 256; ++t) {
    const int wb = (t >> 6) * 64, lane = t & 63;
    for (int j = 0; j < 4; ++j) {
      const float v = ho[4 * t + j];
      if (t >= 200) bad += v != 7.0f;
      else if (t & 1) { oob_zero += v == 0.0f; oob_skip += v == 7.0f; }
      else bad += v != (float)(1024 + 4 * (wb + 63 - lane) + j);
    }
  }
  printf("lds-dma b128: mismatches=%d  out-of-range lanes: wrote zero=%d kept old=%d (of 400)\n", bad, oob_zero, oob_skip);
  printf("sample: t=0 -> %g %g %g %g ; t=1 -> %g ; t=2 -> %g ; t=201 -> %g\n", ho[0], ho[1], ho[2], ho[3], ho[4], ho[8], ho[804]);
  return bad != 0;
}
