// micro-benchmark (gfx950): how much VALU issue rate the other waves of a SIMD keep while one wave per SIMD streams
// matrix instructions of a given shape.  Waves 0-3 (one per SIMD) issue MFMAs back to back on independent
// accumulators; waves 4-15 run a dependency-free stream of v_pk_add_f32.
//   MODE 1: v_mfma_f32_16x16x4_f32 (1024 MAC, 32 cycles)      MODE 2: v_mfma_f32_32x32x2_f32 (2048 MAC, 64 cycles)
//   MODE 3: v_mfma_f32_16x16x32_bf16                          MODE 4: v_mfma_f32_32x32x16_bf16
//   MODE 5: v_mfma_f32_4x4x1_16B_f32 (1024 MAC?)              MODE 6: v_mfma_f64_16x16x4_f64
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void kprobe(float* out, uint64_t* cyc, int iters, int mfWaves, int valuWaves) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {
    if (wave >= mfWaves) return;
    float ma = 1.0f + lane, mb = 2.0f;
    bf16x8 va, vb;
    for (int q = 0; q < 8; ++q) { va[q] = (__bf16)(ma + q); vb[q] = (__bf16)(mb - q); }
    f32x4 a4[4]; f32x16 a16[2]; f64x4 d4[4];
    for (int j = 0; j < 4; ++j) { a4[j] = f32x4{1.f * lane, 2.f, 3.f, 4.f}; d4[j] = f64x4{1.0 * lane, 2.0, 3.0, 4.0}; }
    for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) a16[j][q] = (float)(lane + q);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MODE == 1) a4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, a4[j], 0, 0, 0);
          if (MODE == 2) a16[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, a16[j & 1], 0, 0, 0);
          if (MODE == 3) a4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, a4[j], 0, 0, 0);
          if (MODE == 4) a16[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, a16[j & 1], 0, 0, 0);
          if (MODE == 5) a4[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(ma, mb, a4[j], 0, 0, 0);
          if (MODE == 6) d4[j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ma, (double)mb, d4[j], 0, 0, 0);
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    float s = 0;
    for (int j = 0; j < 4; ++j) s += a4[j][0] + a4[j][3] + (float)d4[j][0];
    for (int j = 0; j < 2; ++j) s += a16[j][0] + a16[j][15];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    return;
  }
  if (wave - 4 >= valuWaves) return;
  f32x2 a[16];
  for (int j = 0; j < 16; ++j) a[j] = f32x2{1.f * j, 2.f * lane};
  f32x2 inc = {1.0f, 0.5f};
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(inc));
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0; for (int j = 0; j < 16; ++j) s += a[j].x + a[j].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

static uint64_t h[256 * 16];
template <int MODE>
static void run(float* out, uint64_t* cyc, int mfw, int vw, const char* label) {
  const int iters = 2000;
  hipMemset(cyc, 0, sizeof(h));
  hipLaunchKernelGGL((kprobe<MODE>), dim3(256), dim3(1024), 0, 0, out, cyc, iters, mfw, vw);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mw = 0, mr = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 16; ++w) { const double v = (double)h[b * 16 + w]; if (w < 4) mw = v > mw ? v : mw; else mr = v > mr ? v : mr; }
  printf("%-34s mfma waves=%d valu waves=%2d : %7.2f cycles per MFMA, %6.2f cycles per v_pk_add_f32 and wave (%s)\n", label, mfw, vw,
         mw / iters / 16.0, mr / iters / 16.0, hipGetErrorString(e));
}
int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, sizeof(h));
  run<1>(out, cyc, 0, 12, "v_pk_add_f32 alone");
  run<1>(out, cyc, 4, 0, "f32 16x16x4 alone");   run<1>(out, cyc, 4, 12, "f32 16x16x4 + VALU");
  run<2>(out, cyc, 4, 0, "f32 32x32x2 alone");   run<2>(out, cyc, 4, 12, "f32 32x32x2 + VALU");
  run<5>(out, cyc, 4, 0, "f32 4x4x1 alone");     run<5>(out, cyc, 4, 12, "f32 4x4x1 + VALU");
  run<3>(out, cyc, 4, 0, "bf16 16x16x32 alone"); run<3>(out, cyc, 4, 12, "bf16 16x16x32 + VALU");
  run<4>(out, cyc, 4, 0, "bf16 32x32x16 alone"); run<4>(out, cyc, 4, 12, "bf16 32x32x16 + VALU");
  run<6>(out, cyc, 4, 0, "f64 16x16x4 alone");   run<6>(out, cyc, 4, 12, "f64 16x16x4 + VALU");
  return 0;
}
