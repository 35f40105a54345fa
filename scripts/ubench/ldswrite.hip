// micro-benchmark (gfx950): cycles for W waves of one workgroup to write a 64 KB LUT stage into LDS with
// different store instructions (values change every iteration).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: ds_write2_b32 (rows r, r+1 of a 16x16 tile, 528-byte rows)   1: ds_write_b32 x4
//      2: ds_write_addtid_b32                                           3: ds_write_b64 (image pair per lane)
//      4: ds_write_b128 (four images per lane)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, uint64_t* cyc, int iters, int writers) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= writers) return;
  f32x4 acc = {1.f * lane, 2.f, 3.f, 4.f};
  const int perWave = 65536 / writers;           // bytes of the stage this wave writes per iteration
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t base = (it & 1) * 67584u + wave * (uint32_t)(perWave + perWave / 32);
    for (int off = 0; off < perWave; off += 1024) {   // 1 KB (= 4 dwords per lane) per step
      if (MODE == 0) {
        char* w = lds + base + (off / 1024) * 2112 + (lane >> 4) * 4 * 528 + (lane & 15) * 4 + (off & 0) ;
        asm volatile("ds_write2_b32 %0, %1, %2 offset1:132\n\tds_write2_b32 %0, %3, %4 offset0:8 offset1:140"
                     :: "v"((uint32_t)(uintptr_t)w + 0u), "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]) : "memory");
      } else if (MODE == 1) {
        const uint32_t a = base + (off / 1024) * 2112 + (lane >> 4) * 4 * 528 + (lane & 15) * 4;
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:528\n\tds_write_b32 %0, %3 offset:1056\n\tds_write_b32 %0, %4 offset:1584"
                     :: "v"(a), "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]) : "memory");
      } else if (MODE == 2) {
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                     "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"
                     :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "s"(base + off) : "m0", "memory");
      } else if (MODE == 3) {
        const uint32_t a = base + off + lane * 8;
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:512" :: "v"(a), "v"(f32x2{acc[0], acc[1]}), "v"(f32x2{acc[2], acc[3]}) : "memory");
      } else {
        const uint32_t a = base + off + lane * 16;
        asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(acc) : "memory");
      }
      acc[0] += 1.0f; acc[2] += 2.0f;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = acc[0] + *reinterpret_cast<float*>(lds + lane * 4);
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  uint64_t h[256 * 16];
  const int iters = 1000;
  const char* names[5] = {"ds_write2_b32", "ds_write_b32 x4", "ds_write_addtid_b32", "ds_write_b64", "ds_write_b128"};
  for (int mode = 0; mode < 5; ++mode)
    for (int writers : {1, 2, 4, 8, 16}) {
      hipMemset(cyc, 0, sizeof(h));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 163840, 0, out, cyc, iters, writers);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 163840, 0, out, cyc, iters, writers);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 163840, 0, out, cyc, iters, writers);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 163840, 0, out, cyc, iters, writers);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 163840, 0, out, cyc, iters, writers);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double mx = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < writers; ++w) mx = h[b * 16 + w] > mx ? h[b * 16 + w] : mx;
      printf("%-22s writers=%2d : %7.1f cycles per 64 KB stage = %6.1f B/clk/CU  (%s)\n", names[mode], writers, mx / iters,
             65536.0 / (mx / iters), hipGetErrorString(e));
    }
  return 0;
}
