// micro-benchmark (gfx950): cycles of one s_barrier round of a 16-wave (or 8-wave) workgroup, one workgroup per CU,
// with nothing else in the loop, and with `work` dependent VALU instructions in ONE wave only (skew).
//   hipcc --offload-arch=gfx950 -O3 -o barrier barrier.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ __launch_bounds__(1024) void kbar(uint64_t* cyc, float* out, int iters, int work) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float a = threadIdx.x;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (wave == 0)
      for (int j = 0; j < work; ++j) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a));
    asm volatile("s_barrier" ::: "memory");
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = a;
}
int main() {
  uint64_t* cyc; float* out; static uint64_t h[256 * 16];
  hipMalloc(&cyc, sizeof(h)); hipMalloc(&out, 256 * 1024 * 4);
  for (int threads : {1024, 512, 256})
    for (int work : {0, 100, 400}) {
      hipMemset(cyc, 0, sizeof(h));
      hipLaunchKernelGGL(kbar, dim3(256), dim3(threads), 0, 0, cyc, out, 10000, work);
      hipDeviceSynchronize();
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double mx = 0; for (int i = 0; i < 256 * 16; ++i) mx = (double)h[i] > mx ? (double)h[i] : mx;
      printf("waves=%2d  dependent v_add in wave 0 per round=%3d : %7.1f cycles per barrier round\n", threads / 64, work, mx / 10000);
    }
  return 0;
}
