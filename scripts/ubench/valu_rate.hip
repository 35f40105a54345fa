// micro-benchmark (gfx950, round 5): issue cost of the accumulate instructions of the three look-up forms, with the
// workgroup shape of k_conv_sym8 (8 waves = two per SIMD, one workgroup per CU).  Every wave runs a dependency-free
// stream over 16 accumulators; cycles per wave-instruction from s_memtime, per wave.
//   MODE 0: v_pk_add_f32 (f32 table: two per ds_read_b128)        MODE 1: v_fma_mix_f32 acc += half * 1.0 (fp16 table, f32 sums: four per ds_read_b64)
//   MODE 2: v_pk_add_f16 (fp16 sums: two per ds_read_b64)         MODE 3: v_add_f32      MODE 4: v_xor_b32_sdwa (the offset fix-up)
//   MODE 5: ds_read_b128 alone (conflict-free rows)               MODE 6: ds_read_b64 alone
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void kprobe(float* out, uint64_t* cyc, int iters) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x2 a[16];
  for (int j = 0; j < 16; ++j) a[j] = f32x2{1.f * j, 2.f * lane};
  f32x2 inc = {1.0f, 0.5f};
  uint32_t h = 0x3c003800u, adr = (uint32_t)lane * (MODE == 5 ? 16u : 8u) + (uint32_t)wave * 1024u;
  for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(inc));
      if (MODE == 1) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[j].x), "+v"(h), "+v"(a[j].y));
      if (MODE == 2) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[j].x) : "v"(h));
      if (MODE == 3) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[j].x) : "v"(inc.x));
      if (MODE == 4) asm volatile("v_xor_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[j].x) : "v"(h));
    }
    if (MODE == 5) {
      asm volatile("ds_read_b128 v[100:103], %0\n\tds_read_b128 v[104:107], %0 offset:8192\n\tds_read_b128 v[108:111], %0 offset:16384\n\t"
                   "ds_read_b128 v[112:115], %0 offset:24576\n\ts_waitcnt lgkmcnt(0)" :: "v"(adr) : "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115");
    }
    if (MODE == 6) {
      asm volatile("ds_read_b64 v[100:101], %0\n\tds_read_b64 v[104:105], %0 offset:8192\n\tds_read_b64 v[108:109], %0 offset:16384\n\t"
                   "ds_read_b64 v[112:113], %0 offset:24576\n\ts_waitcnt lgkmcnt(0)" :: "v"(adr) : "v100","v101","v104","v105","v108","v109","v112","v113");
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  float s = 0;
  for (int j = 0; j < 16; ++j) s += a[j].x + a[j].y;
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, uint64_t* cyc, int per) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kprobe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(kprobe<MODE>, dim3(256), dim3(512), 65536, 0, out, cyc, iters);
  hipLaunchKernelGGL(kprobe<MODE>, dim3(256), dim3(512), 65536, 0, out, cyc, iters);
  hipDeviceSynchronize();
  uint64_t h[2048];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 2048; ++i) s += (double)h[i];
  printf("%-58s %6.2f cycles per wave-instruction (8 waves per CU, %d per iteration)\n", name, s / 2048 / iters / per, per);
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8);
  run<0>("v_pk_add_f32 (2 per look-up pair, f32 table)", out, cyc, 16);
  run<1>("v_fma_mix_f32 sum += half (4 per look-up pair, fp16 table)", out, cyc, 32);
  run<2>("v_pk_add_f16 (2 per look-up pair, fp16 sums)", out, cyc, 16);
  run<3>("v_add_f32", out, cyc, 16);
  run<4>("v_xor_b32_sdwa", out, cyc, 16);
  run<5>("ds_read_b128 (4 in flight, then wait)", out, cyc, 4);
  run<6>("ds_read_b64 (4 in flight, then wait)", out, cyc, 4);
  return 0;
}
