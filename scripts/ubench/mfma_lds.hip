// micro-benchmarks (gfx950): cycles per v_mfma_f32_16x16x4_f32 for one wave per SIMD, with and without
// the LDS writes of the results; and LDS write / read instruction costs.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, uint64_t* cyc, int iters, int activeWaves, const float* in) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= activeWaves) return;
  float a = in[lane], b = in[lane + 64];
  f32x4 acc[4];
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
  char* w0 = lds + wave * 8192 + ((lane >> 4) * 4) * 528 + (lane & 15) * 4;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3 || MODE == 2) { a = a * 1.0001f + 1.0f; b = b * 0.9999f + 1.0f; }   // operands change every iteration
    if (MODE == 0 || MODE == 1 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, MODE == 3 ? f32x4{0,0,0,0} : acc[j], 0, 0, 0);
      if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[j], 0, 0, 0);
      }
    }
    if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        char* w = w0 + j * 1056;
        *reinterpret_cast<float*>(w) = acc[j][0];
        *reinterpret_cast<float*>(w + 528) = acc[j][1];
        *reinterpret_cast<float*>(w + 2 * 528) = acc[j][2];
        *reinterpret_cast<float*>(w + 3 * 528) = acc[j][3];
      }
    }
    if (MODE == 2) { for (int j = 0; j < 4; ++j) acc[j] = f32x4{a + j, b + j, a - j, b - j}; }
    asm volatile("" ::: "memory");
  }
  __builtin_amdgcn_s_waitcnt(0);
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0; for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 1024 + threadIdx.x] = s + *reinterpret_cast<float*>(lds + lane * 4);
}

// MODE 10: ds_read_b64 gather pattern: R reads then adds per block; waves configurable
template <int RB>
__global__ __launch_bounds__(1024) void kread(float* out, uint64_t* cyc, int iters, int activeWaves, const uint32_t* offs) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= activeWaves) return;
  for (int i = threadIdx.x; i < 128 * 132; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  f32x2 acc[RB];
  for (int j = 0; j < RB; ++j) acc[j] = f32x2{0, 0};
  uint32_t o[RB];
  for (int j = 0; j < RB; ++j) o[j] = __builtin_amdgcn_readfirstlane(offs[wave * RB + j]);
  const char* base = lds + lane * 8;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f32x2 v[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) v[j] = *reinterpret_cast<const f32x2*>(base + o[j]);
#pragma unroll
    for (int j = 0; j < RB; ++j) acc[j] += v[j];
    __builtin_amdgcn_sched_barrier(0);
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0; for (int j = 0; j < RB; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

int main() {
  float *out, *in; uint64_t* cyc; uint32_t* offs;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 1024); hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&offs, 16 * 16 * 4);
  hipMemset(in, 0, 1024);
  uint32_t ho[256]; for (int i = 0; i < 256; ++i) ho[i] = ((i * 37) % 128) * 528;
  hipMemcpy(offs, ho, sizeof(ho), hipMemcpyHostToDevice);
  uint64_t h[256 * 16];
  const int iters = 2000;
  auto report = [&](const char* name, int waves, double perIter) {
    hipDeviceSynchronize(); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) mx = h[b * 16 + w] > mx ? h[b * 16 + w] : mx;
    printf("%-40s waves=%2d  cycles/iter(max wave)=%8.1f  -> %s\n", name, waves, mx / iters, "");
    (void)perIter;
  };
  for (int waves : {1, 4, 8, 16}) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 140000, 0, out, cyc, iters, waves, in); report("4 dependent-chain mfma (acc += )", waves, 0);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 140000, 0, out, cyc, iters, waves, in); report("8 mfma (4 tiles x 2 ksteps) + 16 ds_write_b32", waves, 0);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 140000, 0, out, cyc, iters, waves, in); report("16 ds_write_b32 only", waves, 0);
  }
  for (int waves : {1, 4, 8, 12, 16}) {
    hipLaunchKernelGGL(kread<8>, dim3(256), dim3(1024), 140000, 0, out, cyc, iters, waves, offs); report("8 ds_read_b64 + 8 pk_add", waves, 0);
    hipLaunchKernelGGL(kread<16>, dim3(256), dim3(1024), 140000, 0, out, cyc, iters, waves, offs); report("16 ds_read_b64 + 16 pk_add", waves, 0);
  }
  return 0;
}
