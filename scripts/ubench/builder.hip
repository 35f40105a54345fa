// micro-benchmark (gfx950): what a builder wave's stage costs and why.  NB builder waves per CU (one workgroup per
// CU), each multiplying and storing 16 result tiles (4 KB each... 64 ds_write of 4 B per lane) per stage; no gather
// waves, no barrier.  Variants separate the matrix instructions from the LDS stores and try other store forms.
//   MODE 0: 32 f32 MFMA only (two independent accumulators alternate)      MODE 1: 64 ds_write_addtid_b32 only
//   MODE 2: both, interleaved as in production (MFMA, 2 stores, MFMA, 2 stores ...)
//   MODE 3: as 2 with ds_write_b32 (address register)                       MODE 4: as 2, M0 written once per 16 stores
//   MODE 5: 16 MFMA + 64 stores (KS = 1)                                     MODE 6: stores first, then the MFMAs
//   hipcc --offload-arch=gfx950 -O3 -o builder builder.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void kb(float* out, uint64_t* cyc, int iters, int nb) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nb) return;
  f32x4 ca = {1.f * lane, 2.f, 3.f, 4.f}, cb = ca, pa = ca, pb = ca;
  float ma = 1.0f + lane, mb = 2.0f;
  const uint32_t vaddr = (uint32_t)lane * 4u + (uint32_t)wave * 8192u;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t m0v = 65536u * (it & 1) + (uint32_t)(wave & 7) * 8192u;
    if (MODE == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(m0v) : "m0", "memory");
#pragma unroll
    for (int n = 0; n < 8; ++n) {                  // 8 pairs of tiles
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+v"(ma), "+v"(mb));          // opaque: the products are not loop invariants
      if (MODE == 6 && n == 0) {
#pragma unroll
        for (int q = 0; q < 32; ++q)
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%3\n\tds_write_addtid_b32 %1 offset:%4"
                       :: "v"(pa[q & 3]), "v"(pb[q & 3]), "s"(m0v), "n"(0), "n"(256) : "m0", "memory");
      }
#define ST2(x, y, o0, o1)                                                                                                  \
  do {                                                                                                                     \
    if (MODE == 1 || MODE == 2 || MODE == 5)                                                                               \
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%3\n\tds_write_addtid_b32 %1 offset:%4"   \
                   :: "v"(x), "v"(y), "s"(m0v), "n"(o0), "n"(o1) : "m0", "memory");                                        \
    if (MODE == 4)                                                                                                         \
      asm volatile("ds_write_addtid_b32 %0 offset:%2\n\tds_write_addtid_b32 %1 offset:%3"                                  \
                   :: "v"(x), "v"(y), "n"(o0), "n"(o1) : "memory");                                                        \
    if (MODE == 3)                                                                                                         \
      asm volatile("ds_write_b32 %2, %0 offset:%3\n\tds_write_b32 %2, %1 offset:%4"                                        \
                   :: "v"(x), "v"(y), "v"(vaddr), "n"(o0), "n"(o1) : "memory");                                            \
  } while (0)
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 1) ca = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, zero, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ST2(pa[0], pa[1], (n & 3) * 2048, (n & 3) * 2048 + 256);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 1) cb = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, zero, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ST2(pa[2], pa[3], (n & 3) * 2048 + 512, (n & 3) * 2048 + 768);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 1 && MODE != 5) ca = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, ca, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ST2(pb[0], pb[1], (n & 3) * 2048 + 1024, (n & 3) * 2048 + 1280);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 1 && MODE != 5) cb = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, cb, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ST2(pb[2], pb[3], (n & 3) * 2048 + 1536, (n & 3) * 2048 + 1792);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 1) { pa = ca; pb = cb; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_waitcnt(0);
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = pa[0] + pb[1] + ca[2] + cb[3];
}

static uint64_t h[256 * 16];
template <int MODE>
static void run(float* out, uint64_t* cyc, int nb, const char* label) {
  const int iters = 1000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kb<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipMemset(cyc, 0, sizeof(h));
  hipLaunchKernelGGL((kb<MODE>), dim3(256), dim3(1024), 131072, 0, out, cyc, iters, nb);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0;
  for (int i = 0; i < 256 * 16; ++i) mx = (double)h[i] > mx ? (double)h[i] : mx;
  printf("%-58s builder waves/CU=%2d : %7.1f cycles per stage (%s)\n", label, nb, mx / iters, hipGetErrorString(e));
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, sizeof(h));
  for (int nb : {4, 8, 16}) {
    run<0>(out, cyc, nb, "32 f32 MFMA");
    run<1>(out, cyc, nb, "64 ds_write_addtid_b32");
    run<2>(out, cyc, nb, "32 MFMA + 64 addtid stores interleaved (production)");
    run<3>(out, cyc, nb, "32 MFMA + 64 ds_write_b32 interleaved");
    run<4>(out, cyc, nb, "32 MFMA + 64 addtid stores, M0 written once");
    run<5>(out, cyc, nb, "16 MFMA + 64 addtid stores (KS = 1)");
    run<6>(out, cyc, nb, "64 addtid stores, then 32 MFMA");
  }
  return 0;
}
