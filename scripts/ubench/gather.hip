// micro-benchmark (gfx950): the production look-up block (gather8: s_bfe, v_mad_u32_u24, ds_read_b64, counted
// waits, in-place v_pk_add_f32) alone, for 4..16 waves, optionally with 4 extra waves writing a 64 KB stage
// with ds_write2_b32 or ds_write_addtid_b32 at the same time.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
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
// 16 look-ups in flight: temporaries are FIXED physical registers named in the clobber list (an asm statement
// may have at most 30 operands): values v[96:127], addresses v[88:95] (reused for the second half as soon as the
// first eight reads are issued), scalars s[84:91]
#define G16_X4(w, t0, t1, t2, t3)                                                                 \
  "s_and_b32 " t0 ", %[" w "], 0xff\n\ts_bfe_u32 " t1 ", %[" w "], 0x80008\n\t"                 \
  "s_bfe_u32 " t2 ", %[" w "], 0x80010\n\ts_lshr_b32 " t3 ", %[" w "], 24\n\t"
#define G16_MAD(a, t) "v_mad_u32_u24 " a ", " t ", %[rb], %[b]\n\t"
#define G16_RD(v, a) "ds_read_b64 " v ", " a "\n\t"
#define G16_ACC(n, c, v) "s_waitcnt lgkmcnt(" n ")\n\tv_pk_add_f32 %[" c "], " v ", %[" c "]\n\t"
__device__ __forceinline__ void gather16(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t base,
                                         uint32_t rowb, int valid) {
  asm volatile("s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lqskip%=\n\t"
               G16_X4("w0", "s84", "s85", "s86", "s87") G16_X4("w1", "s88", "s89", "s90", "s91")
               G16_MAD("v88", "s84") G16_MAD("v89", "s85") G16_MAD("v90", "s86") G16_MAD("v91", "s87")
               G16_MAD("v92", "s88") G16_MAD("v93", "s89") G16_MAD("v94", "s90") G16_MAD("v95", "s91")
               G16_RD("v[96:97]", "v88") G16_RD("v[98:99]", "v89") G16_RD("v[100:101]", "v90") G16_RD("v[102:103]", "v91")
               G16_RD("v[104:105]", "v92") G16_RD("v[106:107]", "v93") G16_RD("v[108:109]", "v94") G16_RD("v[110:111]", "v95")
               G16_X4("w2", "s84", "s85", "s86", "s87") G16_X4("w3", "s88", "s89", "s90", "s91")
               G16_MAD("v88", "s84") G16_MAD("v89", "s85") G16_MAD("v90", "s86") G16_MAD("v91", "s87")
               G16_MAD("v92", "s88") G16_MAD("v93", "s89") G16_MAD("v94", "s90") G16_MAD("v95", "s91")
               G16_RD("v[112:113]", "v88") G16_RD("v[114:115]", "v89") G16_RD("v[116:117]", "v90") G16_RD("v[118:119]", "v91")
               G16_RD("v[120:121]", "v92") G16_RD("v[122:123]", "v93") G16_RD("v[124:125]", "v94") G16_RD("v[126:127]", "v95")
               G16_ACC("15", "c0", "v[96:97]") G16_ACC("14", "c1", "v[98:99]") G16_ACC("13", "c2", "v[100:101]") G16_ACC("12", "c3", "v[102:103]")
               G16_ACC("11", "c4", "v[104:105]") G16_ACC("10", "c5", "v[106:107]") G16_ACC("9", "c6", "v[108:109]") G16_ACC("8", "c7", "v[110:111]")
               G16_ACC("7", "c8", "v[112:113]") G16_ACC("6", "c9", "v[114:115]") G16_ACC("5", "c10", "v[116:117]") G16_ACC("4", "c11", "v[118:119]")
               G16_ACC("3", "c12", "v[120:121]") G16_ACC("2", "c13", "v[122:123]") G16_ACC("1", "c14", "v[124:125]") G16_ACC("0", "c15", "v[126:127]")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
                 [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7]), [c8] "+v"(acc[8]), [c9] "+v"(acc[9]),
                 [c10] "+v"(acc[10]), [c11] "+v"(acc[11]), [c12] "+v"(acc[12]), [c13] "+v"(acc[13]), [c14] "+v"(acc[14]),
                 [c15] "+v"(acc[15])
               : [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2), [w3] "s"(w3), [b] "v"(base), [rb] "v"(rowb), [ok] "s"(valid)
               : "scc", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "v88", "v89", "v90", "v91", "v92", "v93", "v94",
                 "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108",
                 "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",
                 "v122", "v123", "v124", "v125", "v126", "v127");
}
// variant: 16 reads in flight (two index words pairs), waits only twice
// WRITERS: 0 none, 1 four waves ds_write2_b32 (64 KB per iteration in total), 2 four waves ds_write_addtid_b32
template <int WRITERS, int BAR, int MF, int G16 = 0>
__global__ __launch_bounds__(1024) void kg(float* out, uint64_t* cyc, int iters, int readers, const uint32_t* idx, int rblocks = 4, int mfhalf = 0) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  if (wave < 4) {
    if (WRITERS == 0) return;
    uint64_t t0 = __builtin_readcyclecounter();
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
    float ma = 1.0f + lane, mb = 2.0f;
    for (int it = 0; it < iters; ++it) {
      // one stage = 16 tiles x 4 dwords per lane for this wave (64 KB / 4 waves)
      for (int tl = 0; tl < 16; tl += 4) {
        if (MF == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc[j], 0, 0, 0);
            if (!mfhalf) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, acc[j], 0, 0, 0);
          }
        }
        if (MF == 3) {   // three bf16 16x16x32 per tile (exact 3-way split of both operands) + ~6 VALU per tile of operand prep
          typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
          bf16x8 va, vb, vc;
          for (int q = 0; q < 8; ++q) { va[q] = (__bf16)(ma + q); vb[q] = (__bf16)(mb - q); vc[q] = (__bf16)(ma * 0.5f + q); }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vc, acc[j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vc, vb, acc[j], 0, 0, 0);
          ma = ma * 1.0001f + 0.5f; mb = mb * 0.9999f + 0.25f;
        }
        if (MF == 2) {   // one bf16 16x16x32 per tile (hi/lo split operands) instead of two f32 16x16x4
          typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
          bf16x8 va, vb;
          for (int q = 0; q < 8; ++q) { va[q] = (__bf16)(ma + q); vb[q] = (__bf16)(mb - q); }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[j], 0, 0, 0);
        }
        if (WRITERS == 1) {
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
          const uint32_t m0v = 66048u * (it & 1) + wave * 2 * 8256 + tl * 1024;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            asm volatile("s_mov_b32 m0, %4\n\tds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                         "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"
                         :: "v"(acc[j][0]), "v"(acc[j][1]), "v"(acc[j][2]), "v"(acc[j][3]), "s"(m0v + j * 1024 * 0 + (j & 1) * 8256) : "m0", "memory");
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
  uint32_t w[8];
  for (int j = 0; j < 8; ++j) w[j] = __builtin_amdgcn_readfirstlane(idx[wave * 8 + j]);
  uint32_t rowb; asm volatile("v_mov_b32 %0, 0x210" : "=v"(rowb));
  const uint32_t base = lane * 8;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t st = base + 67584u * (it & 1);
    const int nblk = __builtin_amdgcn_readfirstlane((wave - 4 >= readers) ? 0 : 1);
    if (G16) {
      gather16(&acc[0], w[0], w[1], w[2], w[3], st, rowb, nblk);
      gather16(&acc[16], w[4], w[5], w[6], w[7], st, rowb, __builtin_amdgcn_readfirstlane(nblk && rblocks > 2));
      if (BAR) asm volatile("s_barrier" ::: "memory");
      continue;
    }
    gather8(&acc[0], w[0], w[1], st, rowb, nblk);
    gather8(&acc[8], w[2], w[3], st, rowb, __builtin_amdgcn_readfirstlane(nblk && rblocks > 1));
    gather8(&acc[16], w[4], w[5], st, rowb, __builtin_amdgcn_readfirstlane(nblk && rblocks > 2));
    gather8(&acc[24], w[6], w[7], st, rowb, __builtin_amdgcn_readfirstlane(nblk && rblocks > 3));
    if (BAR) asm volatile("s_barrier" ::: "memory");
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0; for (int j = 0; j < 32; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}

int main() {
  float* out; uint64_t* cyc; uint32_t* idx;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&idx, 16 * 8 * 4);
  uint32_t hi[128]; for (int i = 0; i < 128; ++i) { uint32_t v = 0; for (int b = 0; b < 4; ++b) v |= ((i * 29 + b * 37 + 5) % 128) << (8 * b); hi[i] = v; }
  hipMemcpy(idx, hi, sizeof(hi), hipMemcpyHostToDevice);
  uint64_t h[256 * 16];
  const int iters = 1000;
  auto report = [&](const char* name, int readers) {
    hipDeviceSynchronize(); hipMemset(cyc, 0, sizeof(h)); };
  (void)report;
  for (int mode = 0; mode < 11; ++mode)
    for (int readers : {0, 4, 8, 12}) {
      if (mode == 0 && readers == 0) continue;
      if (mode >= 3 && readers != 12) continue;
      hipMemset(cyc, 0, sizeof(h));
      if (mode == 0) hipLaunchKernelGGL((kg<0, 0, 0>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 1) hipLaunchKernelGGL((kg<1, 0, 0>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 2) hipLaunchKernelGGL((kg<2, 0, 0>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 3) hipLaunchKernelGGL((kg<1, 1, 0>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 4) hipLaunchKernelGGL((kg<1, 0, 1>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 5) hipLaunchKernelGGL((kg<1, 1, 1>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 6) hipLaunchKernelGGL((kg<2, 1, 1>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 7) hipLaunchKernelGGL((kg<1, 1, 2>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 8) hipLaunchKernelGGL((kg<2, 1, 2>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 9) hipLaunchKernelGGL((kg<1, 1, 3>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      if (mode == 10) hipLaunchKernelGGL((kg<2, 1, 3>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, readers, idx);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double mr = 0, mw = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) { double v = (double)h[b * 16 + w]; if (w < 4) mw = v > mw ? v : mw; else mr = v > mr ? v : mr; }
      printf("writers=%s readers=%2d : reader cycles per 32 look-ups (slowest wave) %7.1f   writer cycles per 64KB stage %7.1f  (%s)\n",
             mode == 0 ? "none  " : (mode == 1 ? "write2" : (mode == 2 ? "addtid" : (mode == 3 ? "write2+barrier" : (mode == 4 ? "write2+mfma" : (mode == 5 ? "write2+mfma+barrier" : (mode == 6 ? "addtid+mfma+barrier" : (mode == 7 ? "write2+bf16mfma+barrier" : (mode == 8 ? "addtid+bf16mfma+barrier" : (mode == 9 ? "write2+3xbf16mfma+barrier" : "addtid+3xbf16mfma+barrier"))))))))), readers, mr / iters, mw / iters, hipGetErrorString(e));
    }
  for (int g16 = 0; g16 < 2; ++g16)
    for (int cfg = 0; cfg < 3; ++cfg) {
      hipMemset(cyc, 0, sizeof(h));
      const int rb = cfg == 2 ? 2 : 4, half = cfg == 2 ? 1 : 0;
      if (cfg == 0) { if (g16) hipLaunchKernelGGL((kg<0, 0, 0, 1>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, 12, idx, 4, 0);
                      else hipLaunchKernelGGL((kg<0, 0, 0, 0>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, 12, idx, 4, 0); }
      else { if (g16) hipLaunchKernelGGL((kg<1, 1, 1, 1>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, 12, idx, rb, half);
             else hipLaunchKernelGGL((kg<1, 1, 1, 0>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, 12, idx, rb, half); }
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double mr = 0;
      for (int b = 0; b < 256; ++b) for (int w = 4; w < 16; ++w) { double v = (double)h[b * 16 + w]; mr = v > mr ? v : mr; }
      printf("%s look-up blocks, %s: %7.1f cycles per stage (%s)\n", g16 ? "16-deep" : " 8-deep",
             cfg == 0 ? "12 readers alone, 32 look-ups" : (cfg == 1 ? "32 look-ups + write2 + 32 MFMA + barrier" : "16 look-ups + write2 + 16 MFMA + barrier"),
             mr / iters, hipGetErrorString(e));
    }
  // conv1-like stage: 12 readers x 2 blocks (16 look-ups), writers 16 f32 MFMA + 64 KB, barrier
  for (int rb : {1, 2, 3, 4})
    for (int half : {1, 0}) {
      hipMemset(cyc, 0, sizeof(h));
      hipLaunchKernelGGL((kg<1, 1, 1>), dim3(256), dim3(1024), 163840, 0, out, cyc, iters, 12, idx, rb, half);
      hipDeviceSynchronize();
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double mr = 0;
      for (int b = 0; b < 256; ++b) for (int w = 4; w < 16; ++w) { double v = (double)h[b * 16 + w]; mr = v > mr ? v : mr; }
      printf("write2+mfma+barrier, 12 readers x %d blocks of 8 look-ups, %d f32 MFMA per writer: %7.1f cycles per stage\n", rb, half ? 16 : 32, mr / iters);
    }
  return 0;
}
