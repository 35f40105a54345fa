// mfma_clock.hip — what does a CU sustain in v_mfma_f32_16x16x4_f32 when all four SIMDs stream them, and at which shader
// clock?  (clock64 = shader cycles, wall_clock64 = 100 MHz constant.)  One 1024-thread workgroup per CU, `waves` of its 16
// waves issue `iters` x 18 independent MFMAs; reports cycles per MFMA and SIMD, and the clock the kernel ran at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void kmfma(float* out, uint64_t* cyc, uint64_t* wall, int iters, int waves) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 acc[18];
  for (int i = 0; i < 18; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + lane, b = 0.5f;
  __syncthreads();
  const uint64_t t0 = clock64(), w0 = wall_clock64();
  if (wave < waves) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 18; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      a += 1.0f;
    }
  }
  __syncthreads();
  const uint64_t t1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
  float s = 0;
  for (int i = 0; i < 18; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

int main() {
  float* out; uint64_t *cyc, *wall;
  hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&cyc, 512 * 8); hipMalloc(&wall, 512 * 8);
  uint64_t hc[512], hw[512];
  for (int rep = 0; rep < 2; ++rep)
    for (int waves : {4, 8, 16}) {
      for (int blocks : {256, 32}) {
        const int iters = 20000 / (waves / 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kmfma, dim3(blocks), dim3(1024), 0, 0, out, cyc, wall, iters, waves);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc, cyc, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(hw, wall, blocks * 8, hipMemcpyDeviceToHost);
        double c = 0, w = 0;
        for (int i = 0; i < blocks; ++i) { c += hc[i]; w += hw[i]; }
        c /= blocks; w /= blocks;
        const double mfmaPerSimd = (double)iters * 18 * waves / 4;
        printf("blocks %3d waves %2d: %.3f ms, %.1f shader cycles per MFMA and SIMD, shader clock %.0f MHz (100 MHz wall), %.1f TFLOP/s\n",
               blocks, waves, ms, c / mfmaPerSimd, c / w * 100.0, mfmaPerSimd * 4 * blocks * 2048 / (ms * 1e-3) / 1e12);
      }
    }
  return 0;
}
