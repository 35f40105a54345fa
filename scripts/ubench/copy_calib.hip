// calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM: calibrate on a known byte
// count in your own access pattern): streaming copies of N bytes with 4, 8 and 16 bytes per lane, the access widths
// of the conv/FC kernels (builder operand loads: 4 B per lane; epilogue stores: 16 B per lane; glue kernels: 16 B).
//   hipcc --offload-arch=gfx950 -O3 -o copy_calib copy_calib.hip ; rocprofv3 --pmc FETCH_SIZE -- ./copy_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename T>
__global__ void k_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// read-only (sum) and write-only variants separate the two directions
template <typename T>
__global__ void k_read(const T* __restrict__ src, float* __restrict__ out, size_t n) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T v = src[i];
    s += reinterpret_cast<const float*>(&v)[0];
  }
  if (s == 12345.678f) out[0] = s;
}
template <typename T>
__global__ void k_write(T* __restrict__ dst, size_t n) {
  T v;
  for (unsigned j = 0; j < sizeof(T) / 4; ++j) reinterpret_cast<float*>(&v)[j] = 1.0f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
int main() {
  const size_t bytes = (size_t)2 << 30;   // 2 GiB per direction: far beyond the 256 MiB Infinity Cache
  char *a, *b; float* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  const dim3 grid(256 * 16), blk(256);
  hipLaunchKernelGGL(k_read<float>, grid, blk, 0, 0, (const float*)a, o, bytes / 4);
  hipLaunchKernelGGL(k_read<float2>, grid, blk, 0, 0, (const float2*)a, o, bytes / 8);
  hipLaunchKernelGGL(k_read<float4>, grid, blk, 0, 0, (const float4*)a, o, bytes / 16);
  hipLaunchKernelGGL(k_write<float>, grid, blk, 0, 0, (float*)b, bytes / 4);
  hipLaunchKernelGGL(k_write<float2>, grid, blk, 0, 0, (float2*)b, bytes / 8);
  hipLaunchKernelGGL(k_write<float4>, grid, blk, 0, 0, (float4*)b, bytes / 16);
  hipLaunchKernelGGL(k_copy<float4>, grid, blk, 0, 0, (const float4*)a, (float4*)b, bytes / 16);
  hipDeviceSynchronize();
  printf("each kernel moves %zu bytes per direction\n", bytes);
  return 0;
}
