// micro-benchmark: host-to-device rate of one 1000-image fp32 batch (618 MB) by the kind of host memory.
//   hipcc --offload-arch=gfx950 -O2 -o h2d_probe h2d_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double run(void* dst, const void* src, size_t bytes, hipStream_t st) {
  hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 4; ++i) hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  hipStreamSynchronize(st);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 4;
}
int main() {
  const size_t bytes = (size_t)1000 * 3 * 227 * 227 * 4;
  void* dev; hipMalloc(&dev, bytes);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  struct { const char* name; unsigned flags; } kinds[] = {{"hipHostMalloc default", hipHostMallocDefault}, {"hipHostMalloc portable", hipHostMallocPortable},
      {"hipHostMalloc numa-user", hipHostMallocNumaUser}, {"hipHostMalloc non-coherent", hipHostMallocNonCoherent},
      {"hipHostMalloc portable|non-coherent", hipHostMallocPortable | hipHostMallocNonCoherent}};
  for (auto& k : kinds) {
    void* h = nullptr;
    if (hipHostMalloc(&h, bytes, k.flags) != hipSuccess) { printf("%-40s alloc failed\n", k.name); (void)hipGetLastError(); continue; }
    memset(h, 1, bytes);
    const double t = run(dev, h, bytes, st);
    printf("%-40s %.2f ms  %.1f GB/s\n", k.name, t * 1e3, bytes / t / 1e9);
    hipHostFree(h);
  }
  {
    void* h = malloc(bytes); memset(h, 1, bytes);
    double t = run(dev, h, bytes, st);
    printf("%-40s %.2f ms  %.1f GB/s\n", "pageable (malloc)", t * 1e3, bytes / t / 1e9);
    for (unsigned f : {(unsigned)hipHostRegisterDefault, (unsigned)hipHostRegisterPortable}) {
      if (hipHostRegister(h, bytes, f) == hipSuccess) {
        t = run(dev, h, bytes, st);
        printf("hipHostRegister flags %u %*s %.2f ms  %.1f GB/s\n", f, 16, "", t * 1e3, bytes / t / 1e9);
        hipHostUnregister(h);
      } else { printf("register %u failed\n", f); (void)hipGetLastError(); }
    }
    free(h);
  }
  return 0;
}
