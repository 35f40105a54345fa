// probe (gfx950): where does ds_write_addtid_b32 land?  address = M0[?:0] + 16-bit offset + 4 * lane.
//   hipcc --offload-arch=gfx950 -O3 -o addtid_probe addtid_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t m0a, uint32_t m0b, uint32_t m0c, int nop) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) lds[i] = 0;
  __syncthreads();
  uint32_t va = 0xA0000000u + threadIdx.x, vb = 0xB0000000u + threadIdx.x, vc = 0xC0000000u + threadIdx.x,
           vd = 0xD0000000u + threadIdx.x;
  if (nop) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:0" ::"s"(m0a), "v"(va) : "m0", "memory");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:65532" ::"s"(m0b), "v"(vb) : "m0", "memory");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:0" ::"s"(m0c), "v"(vc) : "m0", "memory");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:256" ::"s"(m0a), "v"(vd) : "m0", "memory");
  } else {
    asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:0" ::"s"(m0a), "v"(va) : "m0", "memory");
    asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:65532" ::"s"(m0b), "v"(vb) : "m0", "memory");
    asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:0" ::"s"(m0c), "v"(vc) : "m0", "memory");
    asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:256" ::"s"(m0a), "v"(vd) : "m0", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t* out; hipMalloc(&out, 160 * 1024);
  static uint32_t h[160 * 1024 / 4];
  for (int nop = 0; nop < 2; ++nop) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, out, 0x100u, 65284u, 0x12340u, nop);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("nop=%d (%s): expected A at 0x100, B at %u (0x%x), C at 0x12340 if M0 is wider than 16 bits (0x2340 if not), D at 0x200\n",
           nop, hipGetErrorString(e), 65284u + 65532u, 65284u + 65532u);
    for (int i = 0; i < 160 * 1024 / 4; ++i)
      if (h[i] && (h[i] & 63) == 0) printf("  tag %c lane0 at byte 0x%x (%d)\n", "ABCD"[(h[i] >> 28) - 10], i * 4, i * 4);
  }
  return 0;
}
