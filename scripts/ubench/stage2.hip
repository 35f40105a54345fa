// micro-benchmark (gfx950), round 2: one LUT stage period of the conv/FC kernels with the candidate instruction
// streams, one 16-wave workgroup per CU (4 builder waves + 12 gather waves), cycles per stage by s_memtime.
//
//   gather  R64  : production block (8 x ds_read_b64 at a wave-uniform row, lane = image pair, rows of 528 B)
//           R128 : 8 x ds_read_b128, lane = four images, lanes 0-31 read the row of one output channel and lanes
//                  32-63 the row of another (two look-ups per LDS instruction); addresses by v_add_u32_sdwa from
//                  packed pre-scaled u16 row offsets; stage layout [image tile][row][16 images] (tile stride 8256 B)
//   builder W32  : ds_write_b32 of the MFMA result registers into 528-byte rows (production)
//           WADD : ds_write_addtid_b32 into the tile layout
//   MFMA         : n x v_mfma_f32_16x16x4_f32 per builder and stage (32 = conv3..5 / VGG, 16 = conv1)
// plus three probes: f32 MFMA beside VALU-only waves on the same SIMD, bf16 16x16x32 issue rate, v_pk_add_f32 rate.
//   hipcc --offload-arch=gfx950 -O3 -o stage2 stage2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define QCNN_X4(w, t0, t1, t2, t3)                                                                         \
  "s_and_b32 %[" t0 "], %[" w "], 0xff\n\ts_bfe_u32 %[" t1 "], %[" w "], 0x80008\n\t"                    \
  "s_bfe_u32 %[" t2 "], %[" w "], 0x80010\n\ts_lshr_b32 %[" t3 "], %[" w "], 24\n\t"
#define QCNN_MAD(a, t) "v_mad_u32_u24 %[" a "], %[" t "], %[rb], %[b]\n\t"
#define QCNN_RD(v, a) "ds_read_b64 %[" v "], %[" a "]\n\t"
#define QCNN_ACC(n, c, v) "s_waitcnt lgkmcnt(" n ")\n\tv_pk_add_f32 %[" c "], %[" v "], %[" c "]\n\t"
__device__ __forceinline__ void gather8(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t base, uint32_t rowb, int valid) {
  f32x2 v0, v1, v2, v3, v4, v5, v6, v7;
  uint32_t a0, a1, a2, a3, a4, a5, a6, a7, t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile("s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lqskip%=\n\t"
               QCNN_X4("w0", "t0", "t1", "t2", "t3") QCNN_X4("w1", "t4", "t5", "t6", "t7")
               QCNN_MAD("a0", "t0") QCNN_MAD("a1", "t1") QCNN_MAD("a2", "t2") QCNN_MAD("a3", "t3")
               QCNN_MAD("a4", "t4") QCNN_MAD("a5", "t5") QCNN_MAD("a6", "t6") QCNN_MAD("a7", "t7")
               QCNN_RD("v0", "a0") QCNN_RD("v1", "a1") QCNN_RD("v2", "a2") QCNN_RD("v3", "a3")
               QCNN_RD("v4", "a4") QCNN_RD("v5", "a5") QCNN_RD("v6", "a6") QCNN_RD("v7", "a7")
               QCNN_ACC("7", "c0", "v0") QCNN_ACC("6", "c1", "v1") QCNN_ACC("5", "c2", "v2") QCNN_ACC("4", "c3", "v3")
               QCNN_ACC("3", "c4", "v4") QCNN_ACC("2", "c5", "v5") QCNN_ACC("1", "c6", "v6") QCNN_ACC("0", "c7", "v7")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
                 [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7]), [v0] "=&v"(v0), [v1] "=&v"(v1),
                 [v2] "=&v"(v2), [v3] "=&v"(v3), [v4] "=&v"(v4), [v5] "=&v"(v5), [v6] "=&v"(v6), [v7] "=&v"(v7),
                 [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5),
                 [a6] "=&v"(a6), [a7] "=&v"(a7), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3),
                 [t4] "=&s"(t4), [t5] "=&s"(t5), [t6] "=&s"(t6), [t7] "=&s"(t7)
               : [w0] "s"(w0), [w1] "s"(w1), [b] "v"(base), [rb] "v"(rowb), [ok] "s"(valid)
               : "scc");
}

// 8 x ds_read_b128 = 16 look-ups: fixed physical temporaries v[88:95] (addresses), v[96:127] (values)
#define Q_AD(a, w, sel) "v_add_u32_sdwa " a ", %[" w "], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" sel " src1_sel:DWORD\n\t"
#define Q_RD(v, a) "ds_read_b128 " v ", " a "\n\t"
#define Q_ACC(n, c0, c1, lo, hi) \
  "s_waitcnt lgkmcnt(" n ")\n\tv_pk_add_f32 %[" c0 "], " lo ", %[" c0 "]\n\tv_pk_add_f32 %[" c1 "], " hi ", %[" c1 "]\n\t"
__device__ __forceinline__ void gatherq8(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t base,
                                         int validIn) {
  int valid;
  asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(valid) : "v"(validIn));
  asm volatile("s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lqskip%=\n\t"
               Q_AD("v88", "w0", "WORD_0") Q_AD("v89", "w0", "WORD_1") Q_AD("v90", "w1", "WORD_0") Q_AD("v91", "w1", "WORD_1")
               Q_AD("v92", "w2", "WORD_0") Q_AD("v93", "w2", "WORD_1") Q_AD("v94", "w3", "WORD_0") Q_AD("v95", "w3", "WORD_1")
               Q_RD("v[96:99]", "v88") Q_RD("v[100:103]", "v89") Q_RD("v[104:107]", "v90") Q_RD("v[108:111]", "v91")
               Q_RD("v[112:115]", "v92") Q_RD("v[116:119]", "v93") Q_RD("v[120:123]", "v94") Q_RD("v[124:127]", "v95")
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
               : "scc", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",
                 "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114",
                 "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

// four reads = 8 look-ups, temporaries v[112:127] (address in the first register of its destination quad)
#define Q4_AD(a, w, sel) "v_xor_b32_sdwa " a ", %[" w "], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" sel " src1_sel:DWORD\n\t"
__device__ __forceinline__ void gatherq4(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t base, int validIn) {
  int valid;
  asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(valid) : "v"(validIn));
  asm volatile("s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lqskip%=\n\t"
               Q4_AD("v112", "w0", "WORD_0") Q4_AD("v116", "w0", "WORD_1") Q4_AD("v120", "w1", "WORD_0") Q4_AD("v124", "w1", "WORD_1")
               Q_RD("v[112:115]", "v112") Q_RD("v[116:119]", "v116") Q_RD("v[120:123]", "v120") Q_RD("v[124:127]", "v124")
               Q_ACC("3", "c0", "c1", "v[112:113]", "v[114:115]") Q_ACC("2", "c2", "c3", "v[116:117]", "v[118:119]")
               Q_ACC("1", "c4", "c5", "v[120:121]", "v[122:123]") Q_ACC("0", "c6", "c7", "v[124:125]", "v[126:127]")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
                 [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7])
               : [w0] "v"(w0), [w1] "v"(w1), [b] "v"(base), [ok] "s"(valid)
               : "scc", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
                 "v124", "v125", "v126", "v127");
}

// NB builder waves (4 or 8: waves 0..NB-1) write 64 KB / NB each with ds_write_addtid_b32 and issue nmf * 4 / NB f32 MFMA
// each; the other 16 - NB waves gather `rblocks` blocks of 8 look-ups (4 x ds_read_b128) each
template <int NB, int M0MODE = 0>
__global__ __launch_bounds__(1024) void kroles(float* out, uint64_t* cyc, int iters, const uint32_t* idx, int rblocks, int nmf) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  if (wave < NB) {
    uint64_t t0 = __builtin_readcyclecounter();
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
    float ma = 1.0f + lane, mb = 2.0f;
    const int groups = 16 / NB;            // groups of four tiles per stage for this wave
    const int mfPerGroup = nmf / 4;        // per group of four tiles: 8 = two per tile, 4 = one per tile
    for (int it = 0; it < iters; ++it) {
      if (M0MODE == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(65536u * (it & 1) + wave * groups * 4096) : "m0", "memory");
#pragma unroll
      for (int tl = 0; tl < groups; ++tl) {
        if (mfPerGroup >= 4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[j], 0, 0, 0);
        }
        if (mfPerGroup >= 8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, acc[j], 0, 0, 0);
        }
        const uint32_t m0v = 65536u * (it & 1) + (wave * groups + tl) * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (M0MODE == 0)
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                         "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"
                         :: "v"(acc[j][0]), "v"(acc[j][1]), "v"(acc[j][2]), "v"(acc[j][3]), "s"(m0v + j * 1024) : "m0", "memory");
          else
            asm volatile("ds_write_addtid_b32 %0 offset:%4\n\tds_write_addtid_b32 %1 offset:%5\n\t"
                         "ds_write_addtid_b32 %2 offset:%6\n\tds_write_addtid_b32 %3 offset:%7"
                         :: "v"(acc[j][0]), "v"(acc[j][1]), "v"(acc[j][2]), "v"(acc[j][3]), "n"(tl * 4096 + j * 1024),
                            "n"(tl * 4096 + j * 1024 + 256), "n"(tl * 4096 + j * 1024 + 512), "n"(tl * 4096 + j * 1024 + 768) : "memory");
        }
        for (int j = 0; j < 4; ++j) acc[j][0] += 1.0f;
        asm volatile("" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0);
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    return;
  }
  f32x2 acc[48];
  for (int j = 0; j < 48; ++j) acc[j] = f32x2{0, 0};
  uint32_t w[12];
  for (int j = 0; j < 12; ++j) w[j] = idx[256 + ((wave * 2 + (lane >> 5)) * 12 + j) % 700];
  const uint32_t base = ((lane & 31) >> 2) * 8192 | ((lane & 31) >> 3) * 64 | (lane & 3) * 16;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t st = base | 65536u * (it & 1);
#pragma unroll
    for (int b = 0; b < 6; ++b) gatherq4(&acc[8 * b], w[2 * b], w[2 * b + 1], st, __builtin_amdgcn_readfirstlane(b < rblocks));
    asm volatile("s_barrier" ::: "memory");
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0; for (int j = 0; j < 48; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

// symmetric stage: every one of the 16 waves stores 4 result tiles (16 x ds_write_addtid_b32) of the next stage,
// issues nmfw f32 MFMA (4 = KS 1, 8 = KS 2) for the stage after it, and gathers rblocks x 8 look-ups of the current one
template <int ORDER>
__global__ __launch_bounds__(1024) void ksym(float* out, uint64_t* cyc, int iters, const uint32_t* idx, int rblocks, int nmfw) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  f32x4 d[4];
  for (int j = 0; j < 4; ++j) d[j] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
  float ma = 1.0f + lane, mb = 2.0f;
  f32x2 acc[32];
  for (int j = 0; j < 32; ++j) acc[j] = f32x2{0, 0};
  uint32_t w[8];
  for (int j = 0; j < 8; ++j) w[j] = idx[256 + ((wave * 2 + (lane >> 5)) * 8 + j) % 700];
  const uint32_t base = ((lane & 31) >> 2) * 8192 | ((lane & 31) >> 3) * 64 | (lane & 3) * 16;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t st = base | 65536u * (it & 1);
    const uint32_t m0v = 65536u * ((it + 1) & 1) + wave * 4096;
    auto stores = [&]() {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(m0v) : "m0", "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("ds_write_addtid_b32 %0 offset:%4\n\tds_write_addtid_b32 %1 offset:%5\n\t"
                     "ds_write_addtid_b32 %2 offset:%6\n\tds_write_addtid_b32 %3 offset:%7"
                     :: "v"(d[j][0]), "v"(d[j][1]), "v"(d[j][2]), "v"(d[j][3]), "n"(j * 1024), "n"(j * 1024 + 256),
                        "n"(j * 1024 + 512), "n"(j * 1024 + 768) : "memory");
    };
    auto mfmas = [&]() {
      if (nmfw >= 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, d[j], 0, 0, 0);
      }
      if (nmfw >= 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, d[j], 0, 0, 0);
      }
    };
    auto gathers = [&]() {
#pragma unroll
      for (int b = 0; b < 4; ++b) gatherq4(&acc[8 * b], w[2 * b], w[2 * b + 1], st, __builtin_amdgcn_readfirstlane(b < rblocks));
    };
    if (ORDER == 0) { stores(); __builtin_amdgcn_sched_barrier(0); mfmas(); __builtin_amdgcn_sched_barrier(0); gathers(); }
    if (ORDER == 1) { gathers(); __builtin_amdgcn_sched_barrier(0); stores(); __builtin_amdgcn_sched_barrier(0); mfmas(); }
    if (ORDER == 2) { stores(); __builtin_amdgcn_sched_barrier(0); gathers(); __builtin_amdgcn_sched_barrier(0); mfmas(); }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0; for (int j = 0; j < 32; ++j) s += acc[j].x + acc[j].y;
  for (int j = 0; j < 4; ++j) s += d[j][0] + d[j][1] + d[j][2] + d[j][3];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

// RD: 0 = R64, 1 = R128;  WR: 0 none, 1 = W32, 2 = WADD;  nmf = f32 MFMA per builder and stage;  rblocks = look-up
// blocks of a gather wave per stage (R64: 8 look-ups each, R128: 16 each)
template <int RD, int WR, int BAR>
__global__ __launch_bounds__(1024) void kstage(float* out, uint64_t* cyc, int iters, int readers, const uint32_t* idx,
                                               int rblocks, int nmf) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  if (wave < 4) {
    if (WR == 0) return;
    uint64_t t0 = __builtin_readcyclecounter();
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
    float ma = 1.0f + lane, mb = 2.0f;
    const int mfPerGroup = nmf / 4;     // MFMA per group of four tiles (8 = two per tile, 4 = one per tile)
    for (int it = 0; it < iters; ++it) {
      for (int tl = 0; tl < 16; tl += 4) {
        if (mfPerGroup >= 4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[j], 0, 0, 0);
        }
        if (mfPerGroup >= 8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, acc[j], 0, 0, 0);
        }
        if (WR == 1) {
          char* w0 = lds + 67584 * (it & 1) + ((lane >> 4) * 4) * 528 + (wave * 32 + (lane & 15)) * 4 + tl * 16 * 528 / 2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            char* w = w0 + j * 8448 / 2;
            *reinterpret_cast<float*>(w) = acc[j][0];
            *reinterpret_cast<float*>(w + 528) = acc[j][1];
            *reinterpret_cast<float*>(w + 2 * 528) = acc[j][2];
            *reinterpret_cast<float*>(w + 3 * 528) = acc[j][3];
          }
        } else {
          const uint32_t m0v = 66048u * (it & 1) + wave * 2 * 8256 + (tl >> 1) * 1024;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            asm volatile("s_mov_b32 m0, %4\n\tds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                         "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"
                         :: "v"(acc[j][0]), "v"(acc[j][1]), "v"(acc[j][2]), "v"(acc[j][3]),
                            "s"(m0v + (j >> 1) * 1024 + (j & 1) * 8256) : "m0", "memory");
          }
        }
        for (int j = 0; j < 4; ++j) acc[j][0] += 1.0f;
        asm volatile("" ::: "memory");
      }
      if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0);
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    return;
  }
  if (wave - 4 >= readers && !BAR) return;
  f32x2 acc[32];
  for (int j = 0; j < 32; ++j) acc[j] = f32x2{0, 0};
  const int on = __builtin_amdgcn_readfirstlane((wave - 4 >= readers) ? 0 : 1);
  if (RD == 0) {
    uint32_t w[8];
    for (int j = 0; j < 8; ++j) w[j] = __builtin_amdgcn_readfirstlane(idx[wave * 8 + j]);
    uint32_t rowb; asm volatile("v_mov_b32 %0, 0x210" : "=v"(rowb));
    const uint32_t base = lane * 8;
    __builtin_amdgcn_s_setprio(2);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      const uint32_t st = base + 67584u * (it & 1);
      gather8(&acc[0], w[0], w[1], st, rowb, __builtin_amdgcn_readfirstlane(on && rblocks > 0));
      gather8(&acc[8], w[2], w[3], st, rowb, __builtin_amdgcn_readfirstlane(on && rblocks > 1));
      gather8(&acc[16], w[4], w[5], st, rowb, __builtin_amdgcn_readfirstlane(on && rblocks > 2));
      gather8(&acc[24], w[6], w[7], st, rowb, __builtin_amdgcn_readfirstlane(on && rblocks > 3));
      if (BAR) asm volatile("s_barrier" ::: "memory");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  } else {
    // packed u16 row offsets (row' * 64), a different set for each half of the wave
    uint32_t w[8];
    for (int j = 0; j < 8; ++j) w[j] = idx[256 + (wave * 2 + (lane >> 5)) * 8 + j];
    const uint32_t base = ((lane & 31) >> 2) * 8256 + (lane & 3) * 16;
    __builtin_amdgcn_s_setprio(2);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      const uint32_t st = base + 66048u * (it & 1);
      gatherq8(&acc[0], w[0], w[1], w[2], w[3], st, on);
      gatherq8(&acc[16], w[4], w[5], w[6], w[7], st, __builtin_amdgcn_readfirstlane(on && rblocks > 1));
      if (BAR) asm volatile("s_barrier" ::: "memory");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  }
  float s = 0; for (int j = 0; j < 32; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

// probe: waves 0..3 (one per SIMD, presumably) issue f32 (MODE 1) or bf16 16x16x32 (MODE 2) MFMAs back to back,
// the other `valuWaves` waves run a chain-free stream of v_pk_add_f32 (PK = 1) or v_add_f32 (PK = 0)
template <int MODE, int PK>
__global__ __launch_bounds__(1024) void kprobe(float* out, uint64_t* cyc, int iters, int mfWaves, int valuWaves) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {
    if (wave >= mfWaves) return;
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
    float ma = 1.0f + lane, mb = 2.0f;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 va, vb;
    for (int q = 0; q < 8; ++q) { va[q] = (__bf16)(ma + q); vb[q] = (__bf16)(mb - q); }
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MODE == 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[j], 0, 0, 0);
          if (MODE == 2) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[j], 0, 0, 0);
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    float s = 0; for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
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
    for (int j = 0; j < 16; ++j) {
      if (PK) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[j]) : "v"(inc));
      else asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[j].x) : "v"(inc.x));
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0; for (int j = 0; j < 16; ++j) s += a[j].x + a[j].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

static uint64_t h[256 * 16];
static void maxes(uint64_t* cyc, double& mw, double& mr) {
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  mw = mr = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 16; ++w) {
      const double v = (double)h[b * 16 + w];
      if (w < 4) mw = v > mw ? v : mw; else mr = v > mr ? v : mr;
    }
}

int main() {
  float* out; uint64_t* cyc; uint32_t* idx;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, sizeof(h)); hipMalloc(&idx, 1024 * 4);
  uint32_t hi[1024];
  for (int i = 0; i < 256; ++i) { uint32_t v = 0; for (int b = 0; b < 4; ++b) v |= ((i * 29 + b * 37 + 5) % 128) << (8 * b); hi[i] = v; }
  for (int i = 256; i < 1024; ++i) { uint32_t v = 0; for (int b = 0; b < 2; ++b) v |= (((i * 29 + b * 37 + 5) % 128) * 64) << (16 * b); hi[i] = v; }
  hipMemcpy(idx, hi, sizeof(hi), hipMemcpyHostToDevice);
  const int iters = 1000;
  double mw, mr;
#define RUN(RD, WR, BAR, readers, rblocks, nmf, label)                                                               \
  do {                                                                                                               \
    hipMemset(cyc, 0, sizeof(h));                                                                                    \
    hipLaunchKernelGGL((kstage<RD, WR, BAR>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx, rblocks, nmf); \
    hipError_t e = hipDeviceSynchronize();                                                                           \
    maxes(cyc, mw, mr);                                                                                              \
    printf("%-58s readers=%2d lookups/wave=%2d mfma=%2d : gather %7.1f  builder %7.1f cycles/stage (%s)\n", label, readers, \
           (RD ? 16 : 8) * rblocks, nmf, mr / iters, mw / iters, hipGetErrorString(e));                              \
  } while (0)
  // gather alone
  for (int r : {4, 8, 12}) RUN(0, 0, 0, r, 4, 0, "R64 alone");
  for (int r : {4, 8, 12}) RUN(1, 0, 0, r, 2, 0, "R128 alone");
  RUN(0, 0, 0, 12, 2, 0, "R64 alone");
  RUN(1, 0, 0, 12, 1, 0, "R128 alone");
  // gather + stores (no barrier: rates of the two streams side by side)
  RUN(0, 1, 0, 12, 4, 0, "R64 + W32");
  RUN(0, 2, 0, 12, 4, 0, "R64 + WADD (layout mismatch is harmless here)");
  RUN(1, 1, 0, 12, 2, 0, "R128 + W32");
  RUN(1, 2, 0, 12, 2, 0, "R128 + WADD");
  RUN(1, 2, 1, 12, 2, 0, "R128 + WADD + barrier");
  // full stage, conv3-like: 32 look-ups per gather wave, 32 MFMA per builder
  RUN(0, 1, 1, 12, 4, 32, "R64 + W32 + MFMA + barrier (production, conv3-like)");
  RUN(0, 2, 1, 12, 4, 32, "R64 + WADD + MFMA + barrier");
  RUN(1, 1, 1, 12, 2, 32, "R128 + W32 + MFMA + barrier");
  RUN(1, 2, 1, 12, 2, 32, "R128 + WADD + MFMA + barrier (candidate, conv3-like)");
  RUN(1, 2, 1, 12, 2, 16, "R128 + WADD + MFMA + barrier (KS=1)");
  RUN(1, 2, 1, 12, 2, 0, "R128 + WADD + barrier, no MFMA");
  // conv1-like: 16 look-ups per gather wave, 16 MFMA per builder
  RUN(0, 1, 1, 12, 2, 16, "R64 + W32 + MFMA + barrier (production, conv1-like)");
  RUN(1, 2, 1, 12, 1, 16, "R128 + WADD + MFMA + barrier (candidate, conv1-like)");
  RUN(1, 2, 1, 12, 1, 0, "R128 + WADD + barrier, no MFMA");
  // role split: 4 builders + 12 gather waves vs 8 + 8 (same total look-ups and the same 64 KB + MFMA per stage)
#define ROLES(NB, rblocks, nmf, label)                                                                               \
  do {                                                                                                               \
    hipMemset(cyc, 0, sizeof(h));                                                                                    \
    hipLaunchKernelGGL((kroles<NB>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, idx, rblocks, nmf);           \
    hipError_t e = hipDeviceSynchronize();                                                                           \
    maxes(cyc, mw, mr);                                                                                              \
    printf("%-58s builders=%d lookups/gather wave=%2d mfma/stage/SIMD=%2d : %7.1f cycles/stage (%s)\n", label, NB,     \
           8 * rblocks, nmf, (mr > mw ? mr : mw) / iters, hipGetErrorString(e));                                      \
  } while (0)
#define SYM(ORDER, rblocks, nmfw, label)                                                                             \
  do {                                                                                                               \
    hipMemset(cyc, 0, sizeof(h));                                                                                    \
    hipLaunchKernelGGL((ksym<ORDER>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, idx, rblocks, nmfw);         \
    hipError_t e = hipDeviceSynchronize();                                                                           \
    maxes(cyc, mw, mr);                                                                                              \
    printf("%-58s order=%d lookups/wave=%2d (x16) mfma/wave=%d : %7.1f cycles/stage (%s)\n", label, ORDER,           \
           8 * rblocks, nmfw, (mr > mw ? mr : mw) / iters, hipGetErrorString(e));                                    \
  } while (0)
  SYM(0, 3, 8, "symmetric, conv3-like (384 look-ups, 32 MFMA/SIMD)");
  SYM(1, 3, 8, "symmetric, conv3-like (384 look-ups, 32 MFMA/SIMD)");
  SYM(2, 3, 8, "symmetric, conv3-like (384 look-ups, 32 MFMA/SIMD)");
  SYM(0, 4, 8, "symmetric, 512 look-ups, 32 MFMA/SIMD");
  SYM(0, 3, 4, "symmetric, 384 look-ups, 16 MFMA/SIMD");
  SYM(0, 2, 4, "symmetric, conv1-like (256 look-ups, 16 MFMA/SIMD)");
  SYM(0, 1, 4, "symmetric, 128 look-ups, 16 MFMA/SIMD");
  SYM(0, 3, 0, "symmetric, 384 look-ups, no MFMA");
  SYM(0, 0, 0, "symmetric, stores + barrier only");
  SYM(0, 3, 0, "symmetric, 384 look-ups, no MFMA");
#define ROLES1(NB, rblocks, nmf, label)                                                                              \
  do {                                                                                                               \
    hipMemset(cyc, 0, sizeof(h));                                                                                    \
    hipLaunchKernelGGL((kroles<NB, 1>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, idx, rblocks, nmf);        \
    hipError_t e = hipDeviceSynchronize();                                                                           \
    maxes(cyc, mw, mr);                                                                                              \
    printf("%-58s builders=%d lookups/gather wave=%2d mfma/stage/SIMD=%2d : %7.1f cycles/stage (%s)\n", label, NB,     \
           8 * rblocks, nmf, (mr > mw ? mr : mw) / iters, hipGetErrorString(e));                                      \
  } while (0)
  ROLES1(4, 4, 32, "M0 once per stage: 4+12, conv3-like (384 look-ups)");
  ROLES1(8, 6, 32, "M0 once per stage: 8+8,  conv3-like (384 look-ups)");
  ROLES1(4, 2, 16, "M0 once per stage: 4+12, conv1-like (192 look-ups)");
  ROLES1(8, 3, 16, "M0 once per stage: 8+8,  conv1-like (192 look-ups)");
  ROLES1(4, 4, 0, "M0 once per stage: 4+12, no MFMA (384 look-ups)");
  ROLES1(8, 6, 0, "M0 once per stage: 8+8,  no MFMA (384 look-ups)");
  ROLES1(4, 0, 0, "M0 once per stage: 4 builders alone");
  ROLES1(8, 0, 0, "M0 once per stage: 8 builders alone");
  ROLES(4, 0, 0, "M0 per 4 stores: 4 builders alone");
  ROLES(4, 4, 32, "4+12, conv3-like (384 look-ups)");
  ROLES(8, 6, 32, "8+8,  conv3-like (384 look-ups)");
  ROLES(4, 4, 16, "4+12, KS=1 (384 look-ups)");
  ROLES(8, 6, 16, "8+8,  KS=1 (384 look-ups)");
  ROLES(4, 2, 16, "4+12, conv1-like (192 look-ups)");
  ROLES(8, 3, 16, "8+8,  conv1-like (192 look-ups)");
  ROLES(4, 4, 0, "4+12, no MFMA (384 look-ups)");
  ROLES(8, 6, 0, "8+8,  no MFMA (384 look-ups)");
  ROLES(8, 4, 32, "8+8,  256 look-ups, 32 MFMA");
  ROLES(8, 2, 16, "8+8,  128 look-ups, 16 MFMA");
  // probes
#define PROBE(MODE, PK, mfw, vw, label)                                                                              \
  do {                                                                                                               \
    hipMemset(cyc, 0, sizeof(h));                                                                                    \
    hipLaunchKernelGGL((kprobe<MODE, PK>), dim3(256), dim3(1024), 0, 0, out, cyc, iters, mfw, vw);                   \
    hipError_t e = hipDeviceSynchronize();                                                                           \
    maxes(cyc, mw, mr);                                                                                              \
    printf("%-58s mfma waves=%d valu waves=%2d : %6.2f cycles per MFMA, %6.2f cycles per VALU op per wave (%s)\n", label, mfw, \
           vw, mw / iters / 16.0, mr / iters / 16.0, hipGetErrorString(e));                                          \
  } while (0)
  PROBE(1, 1, 4, 0, "f32 16x16x4 MFMA alone");
  PROBE(2, 1, 4, 0, "bf16 16x16x32 MFMA alone");
  PROBE(1, 1, 0, 12, "v_pk_add_f32 alone (3 waves per SIMD)");
  PROBE(1, 0, 0, 12, "v_add_f32 alone (3 waves per SIMD)");
  PROBE(1, 1, 0, 4, "v_pk_add_f32 alone (1 wave per SIMD)");
  PROBE(1, 1, 4, 12, "f32 MFMA beside v_pk_add_f32");
  PROBE(2, 1, 4, 12, "bf16 MFMA beside v_pk_add_f32");
  PROBE(1, 0, 4, 12, "f32 MFMA beside v_add_f32");
  return 0;
}
