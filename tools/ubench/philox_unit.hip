// Cost of one weight-synthesis unit (Philox4x32-10 + 2 Box-Muller = 4 normals) per wave, in isolation:
// 1 or 2 waves per SIMD, dependent chain avoided (independent counters), measured with s_memtime and wall clock.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o /tmp/philox_unit tools/ubench/philox_unit.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../bayesian_torch_amd/csrc/bt_device.h"
using namespace bt;

template <int MODE>
__global__ void k(float* out, unsigned long long* ticks, uint32_t seed, int n) {
  RngKey key{seed, 1, 2, 3};
  float acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {  // full unit
      float z[4];
      philox_normal4(key, 5, threadIdx.x + i * 1024, z);
      acc += z[0] + z[1] + z[2] + z[3];
    } else if (MODE == 1) {  // philox only
      uint32_t r[4];
      philox_block(key, 5, threadIdx.x + i * 1024, r);
      acc += __uint_as_float((r[0] ^ r[1] ^ r[2] ^ r[3]) & 0x3FFFFFFFu);
    } else {  // box-muller only
      float z0, z1, z2, z3;
      const uint32_t a = threadIdx.x * 2654435761u + i * 40503u;
      box_muller(a, a ^ 0x9E3779B9u, z0, z1);
      box_muller(a + 77u, a ^ 0x12345u, z2, z3);
      acc += z0 + z1 + z2 + z3;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, int n) {
  float* out;
  unsigned long long* ticks;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&ticks, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, ticks, 1u, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, ticks, 1u, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t;
  hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-12s threads/block=%4d  %8.1f ns/unit/wave (wall)  %8.2f memtime-ticks/unit\n", name, threads, ms * 1e6 / n, (double)t / n);
  hipFree(out), hipFree(ticks);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("unit", threads, 20000);
    run<1>("philox", threads, 20000);
    run<2>("box-muller", threads, 20000);
  }
  return 0;
}
