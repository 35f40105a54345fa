// micro-benchmark (gfx950): ONE 16-wave workgroup per CU on a 128-image panel (production: 4 builder + 12 gather waves,
// 2 x 64 KB stages, one s_barrier per stage for all 16 waves) against TWO independent 8-wave workgroups per CU on
// 64-image panels (2 builder + 6 gather waves, 2 x 32 KB stages each).  Per wave the work of a stage is identical in
// both (a builder multiplies and stores its two image tiles x 8 row tiles: 32 f32 MFMA + 64 ds_write_addtid_b32; a
// gather wave issues `reads` ds_read_b128 = 2 look-ups each and 2 v_pk_add_f32 per read), so
// cycles-per-stage(16) / cycles-per-stage(8) is the throughput ratio of the two organisations.
//   hipcc --offload-arch=gfx950 -O3 -o occ2 occ2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef ADDR_MAD   // address = offset halfword + lane base by v_mad_u32_u16 with op_sel instead of an SDWA xor
#define Q_SEL_WORD_0 "0"
#define Q_SEL_WORD_1 "1"
#define Q_AD(a, w, sel) "v_mad_u32_u16 " a ", %[" w "], 1, %[b] op_sel:[" Q_SEL_##sel ",0,0,0]\n\t"
#else
#define Q_AD(a, w, sel) "v_xor_b32_sdwa " a ", %[" w "], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" #sel " src1_sel:DWORD\n\t"
#endif
#define Q_RD(v, a) "ds_read_b128 " v ", " a "\n\t"
#define Q_ACC(n, c0, c1, lo, hi) \
  "s_waitcnt lgkmcnt(" n ")\n\tv_pk_add_f32 %[" c0 "], " lo ", %[" c0 "]\n\tv_pk_add_f32 %[" c1 "], " hi ", %[" c1 "]\n\t"
__device__ __forceinline__ void gq8(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t base, int valid) {
  asm volatile("s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lqskip%=\n\t"
               Q_AD("v96", "w0", WORD_0) Q_AD("v100", "w0", WORD_1) Q_AD("v104", "w1", WORD_0) Q_AD("v108", "w1", WORD_1)
               Q_AD("v112", "w2", WORD_0) Q_AD("v116", "w2", WORD_1) Q_AD("v120", "w3", WORD_0) Q_AD("v124", "w3", WORD_1)
               Q_RD("v[96:99]", "v96") Q_RD("v[100:103]", "v100") Q_RD("v[104:107]", "v104") Q_RD("v[108:111]", "v108")
               Q_RD("v[112:115]", "v112") Q_RD("v[116:119]", "v116") Q_RD("v[120:123]", "v120") Q_RD("v[124:127]", "v124")
               Q_ACC("7", "c0", "c1", "v[96:97]", "v[98:99]") Q_ACC("6", "c2", "c3", "v[100:101]", "v[102:103]")
               Q_ACC("5", "c4", "c5", "v[104:105]", "v[106:107]") Q_ACC("4", "c6", "c7", "v[108:109]", "v[110:111]")
               Q_ACC("3", "c8", "c9", "v[112:113]", "v[114:115]") Q_ACC("2", "c10", "c11", "v[116:117]", "v[118:119]")
               Q_ACC("1", "c12", "c13", "v[120:121]", "v[122:123]") Q_ACC("0", "c14", "c15", "v[124:125]", "v[126:127]")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
                 [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7]), [c8] "+v"(acc[8]), [c9] "+v"(acc[9]),
                 [c10] "+v"(acc[10]), [c11] "+v"(acc[11]), [c12] "+v"(acc[12]), [c13] "+v"(acc[13]), [c14] "+v"(acc[14]),
                 [c15] "+v"(acc[15])
               : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [b] "v"(base), [ok] "s"(valid)
               : "scc", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108",
                 "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",
                 "v122", "v123", "v124", "v125", "v126", "v127");
}

// lane -> (unit, place) of the four 16-lane groups a ds_read_b128 is serviced in (MI355X_MICROARCH.md, LDS):
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32
__device__ __forceinline__ void lane_unit(int lane, int& unit, int& place) {
  const int l = lane & 31;
  int u, p;
  if (l < 4) { u = 0; p = l; }
  else if (l < 12) { u = 1; p = l - 4; }
  else if (l < 16) { u = 0; p = l - 8; }
  else if (l < 20) { u = 1; p = l - 8; }
  else if (l < 28) { u = 0; p = l - 12; }
  else { u = 1; p = l - 16; }
  unit = u + 2 * (lane >> 5);
  place = p;
}

// NW waves per workgroup, NB of them builders; STAGE = bytes of one stage (NW * 4 KB: 64 KB or 32 KB)
template <int NW, int NB, int KSTEPS, int HALF = 0>
__global__ __launch_bounds__(NW * 64, 16 / NW) void kwg(float* out, uint64_t* cyc, int iters, const uint32_t* idx, int reads8,
                                                          int bar) {
  extern __shared__ char lds[];
  constexpr uint32_t STAGE = NW * 4096u;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < (int)(2 * STAGE / 4); i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  if (wave < NB) {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
    float ma = 1.0f + lane, mb = 2.0f;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int tl = 0; tl < (HALF ? 2 : 4); ++tl) {   // 4 groups of 4 result tiles = 2 image tiles x 8 row tiles (HALF: one image tile)
        const uint32_t m0v = STAGE * (it & 1) + (uint32_t)(2 * wave + (tl >> 1)) * 8192u + (tl & 1) * 4096u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 r = acc[j];
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:256"
                       :: "v"(r[0]), "v"(r[1]), "s"(m0v + j * 1024) : "m0", "memory");
          __builtin_amdgcn_sched_barrier(0);
          if (KSTEPS > 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:512\n\tds_write_addtid_b32 %1 offset:768"
                       :: "v"(r[2]), "v"(r[3]), "s"(m0v + j * 1024) : "m0", "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (bar) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0);
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
    float s = 0; for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    return;
  }
  f32x2 acc[32];
  for (int j = 0; j < 32; ++j) acc[j] = f32x2{0, 0};
  uint32_t w[8];
  uint32_t base;
  if (NW == 16 && !HALF) {       // lanes 0-31 one row, lanes 32-63 another: 8 image tiles x 4 quarters
    const int quad = lane & 31;
    base = (uint32_t)(quad >> 2) * 8192u | (uint32_t)(quad >> 3) * 64u | (uint32_t)(quad & 3) * 16u;
    for (int j = 0; j < 8; ++j) w[j] = idx[((wave * 2 + (lane >> 5)) * 8 + j) % 768];
  } else {              // every 16-lane service group its own row: 4 image tiles x 4 quarters
    int unit, place;
    lane_unit(lane, unit, place);
    base = (uint32_t)(place >> 2) * 8192u | (uint32_t)(place >> 2) * 64u | (uint32_t)(place & 3) * 16u;
    for (int j = 0; j < 8; ++j) w[j] = idx[((wave * 4 + unit) * 8 + j) % 768];
  }
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t st = base | STAGE * (it & 1);
    gq8(&acc[0], w[0], w[1], w[2], w[3], st, __builtin_amdgcn_readfirstlane(reads8 > 0));
    gq8(&acc[16], w[4], w[5], w[6], w[7], st, __builtin_amdgcn_readfirstlane(reads8 > 1));
    if (HALF) gq8(&acc[0], w[1], w[2], w[5], w[6], st, __builtin_amdgcn_readfirstlane(reads8 > 2));
    if (bar) asm volatile("s_barrier" ::: "memory");
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
  float s = 0; for (int j = 0; j < 32; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

static uint64_t h[512 * 16];
template <int NW, int NB, int KSTEPS, int HALF = 0>
static void run(float* out, uint64_t* cyc, const uint32_t* idx, int reads8, int bar, const char* label) {
  const int iters = 1000, blocks = 256 * 16 / NW;
  const size_t shm = 2 * NW * 4096;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kwg<NW, NB, KSTEPS, HALF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kwg<NW, NB, KSTEPS, HALF>, NW * 64, shm);
  hipMemset(cyc, 0, sizeof(h));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kwg<NW, NB, KSTEPS, HALF>), dim3(blocks), dim3(NW * 64), shm, 0, out, cyc, iters, idx, reads8, bar);
  hipEventRecord(e1);
  hipError_t e = hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0, sum = 0;
  for (int i = 0; i < blocks * NW; ++i) { const double v = (double)h[i]; mx = v > mx ? v : mx; sum += v; }
  printf("%-44s waves/WG=%2d builders=%d ksteps=%d reads/gather wave=%2d occupancy=%d WG/CU : %7.1f cycles/stage (max wave), mean %7.1f, "
         "kernel %.3f ms -> %.1f ns per 128-image stage and CU (%s)\n",
         label, NW, NB, KSTEPS, 8 * reads8, occ, mx / iters, sum / (blocks * NW) / iters, ms,
         ms * 1e6 / iters, hipGetErrorString(e));
}

int main() {
  float* out; uint64_t* cyc; uint32_t* idx;
  hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&cyc, sizeof(h)); hipMalloc(&idx, 1024 * 4);
  uint32_t hi[1024];
  for (int i = 0; i < 1024; ++i) { uint32_t v = 0; for (int b = 0; b < 2; ++b) v |= (((i * 29 + b * 37 + 5) % 128) * 64) << (16 * b); hi[i] = v; }
  hipMemcpy(idx, hi, sizeof(hi), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<16, 4, 2>(out, cyc, idx, 2, 1, "conv3-like, one 16-wave WG per CU");
    run<8, 2, 2>(out, cyc, idx, 2, 1, "conv3-like, two 8-wave WGs per CU");
    run<16, 4, 1>(out, cyc, idx, 1, 1, "conv1-like, one 16-wave WG per CU");
    run<8, 2, 1>(out, cyc, idx, 1, 1, "conv1-like, two 8-wave WGs per CU");
    run<16, 4, 2, 1>(out, cyc, idx, 2, 1, "64-image half panel, 16 waves, 16 reads/wave");
    run<16, 4, 2, 1>(out, cyc, idx, 3, 1, "64-image half panel, 16 waves, 24 reads/wave");
    run<16, 4, 1, 1>(out, cyc, idx, 1, 1, "64-image half panel, conv1-like (KS 1, 8 reads)");
    run<16, 4, 1, 1>(out, cyc, idx, 2, 1, "64-image half panel, conv1-like (KS 1, 16 reads)");
    run<16, 4, 2>(out, cyc, idx, 2, 0, "conv3-like, 16 waves, no barrier");
    run<8, 2, 2>(out, cyc, idx, 2, 0, "conv3-like, 8 waves x 2, no barrier");
  }
  return 0;
}
