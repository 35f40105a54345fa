// qcnn_kernels.hip — hand-written gfx950 (CDNA4) kernels of the Quantized-CNN approximate forward pass.
//
// Hot kernels:
//   k_conv_aprx  fused  GetInPdMat (src/CaffeEva.cc:1261-1296)  +  CalcFeatMap_ConvAprx (:760-868)
//   k_fc_aprx    fused  GetInPdMat                              +  CalcFeatMap_FCntAprx (:968-1025)
// plus the two table kernels (k_decode_cbn, k_build_program).  The glue kernels (row a9) live in qcnn_glue.hip, the
// few-image kernels in qcnn_small.hip.
//
// Mapping (see qcnn_kernels.h for the HBM and LDS layouts, DESIGN.md §3 for the measurements behind it): a
// workgroup (16 waves, one per CU) owns one 128-image panel, one tile of output positions and one slice of
// output channels.  The look-up table is never materialised in HBM: it is produced one STAGE at a time in
// LDS — a stage = 128 code-word rows = G = 128/K consecutive sub-spaces of one source pixel (conv) or of the
// input vector (FC) for the 128 images, laid out [8 image tiles][128 rows][16 images].  Stages are double
// buffered, one s_barrier per stage.  The waves are specialised: four BUILDER waves (one per SIMD) multiply
// stage s+1 out (v_mfma_f32_16x16x4_f32, or ordered VALU mul+add in "exact" mode) and store it with
// ds_write_addtid_b32; twelve GATHER waves consume stage s.  A gather lane carries FOUR images; lanes 0-31 of a
// wave work on one half of the wave's output channels and lanes 32-63 on the other half, so that one
// ds_read_b128 performs two look-ups (two rows x 128 images) and two v_pk_add_f32 accumulate them.  Every
// gather wave owns ALL positions of the workgroup's tile and a slice of the channels: in every stage all
// twelve waves have the same number of look-ups.  Stages are visited in (pixel row-major, sub-space
// ascending) order, which for any one output is exactly the reference's (kh, kw, m) summation order
// (:840-863), so with the exact builder conv/FC outputs are bit-identical to the reference.  The row offsets
// of the look-ups (uint16, pre-scaled) arrive a stage ahead: conv layers with K = 128 read them through a per-layer
// "program" table whose row for a stage is DMA-ed into LDS by one wave (QkProgram), the others straight from the
// plain table through the vector memory path.
#include "qcnn_kernels.h"
#include "qcnn_dev.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <queue>
#include <utility>
#include <vector>


namespace {

constexpr int NW = 16;                         // waves per workgroup of the two hot kernels (4 per SIMD)
constexpr int NBW = 4;                         // builder waves (one per SIMD) — MFMA + LDS writes
constexpr int NGW = QCNN_GATHER_WAVES;         // gather waves — LDS reads + packed adds
constexpr int STAGE_ROWS = QCNN_STAGE_ROWS;
// s_setprio of the two wave roles.  Measured (profiles/r2_*/variants.log): any setting with the gather waves
// ABOVE the builders costs 6 % (the builder's store stream is the pole of most stages); equal priorities and
// builder-above-gather measure the same.
#ifndef QCNN_PRIO_BUILDER
#define QCNN_PRIO_BUILDER 0
#endif
#ifndef QCNN_PRIO_GATHER
#define QCNN_PRIO_GATHER 0
#endif
static_assert(NW == NBW + NGW, "wave roles");


#ifdef QCNN_TRACE
// Debug build only (scripts/trace_stage.py): workgroup `qcnn_trace_block` records, for every wave and the first
// 64 stage periods, the cycle at which it arrives at the stage barrier and the cycle at which it leaves it.
__device__ unsigned long long qcnn_trace_buf[16 * 64 * 2 + 16 + 16 * 64];
__device__ int qcnn_trace_block = 0;
#define TR_ARRIVE(s) do { if ((int)blockIdx.x == qcnn_trace_block && blockIdx.y == 0 && blockIdx.z == 0 && (s) < 64 && (threadIdx.x & 63) == 0) \
    qcnn_trace_buf[((threadIdx.x >> 6) * 64 + (s)) * 2] = __builtin_readcyclecounter(); } while (0)
#define TR_LEAVE(s) do { if ((int)blockIdx.x == qcnn_trace_block && blockIdx.y == 0 && blockIdx.z == 0 && (s) < 64 && (threadIdx.x & 63) == 0) \
    qcnn_trace_buf[((threadIdx.x >> 6) * 64 + (s)) * 2 + 1] = __builtin_readcyclecounter(); } while (0)
#define TR_MID(s) do { if ((int)blockIdx.x == qcnn_trace_block && blockIdx.y == 0 && blockIdx.z == 0 && (s) < 64 && (threadIdx.x & 63) == 0) \
    qcnn_trace_buf[16 * 64 * 2 + 16 + (threadIdx.x >> 6) * 64 + (s)] = __builtin_readcyclecounter(); } while (0)
#define TR_ROLE(r) do { if ((int)blockIdx.x == qcnn_trace_block && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0) \
    qcnn_trace_buf[16 * 64 * 2 + (threadIdx.x >> 6)] = (r).builder ? 100 + (r).idx : (r).idx; } while (0)
#else
#define TR_ARRIVE(s) do {} while (0)
#define TR_MID(s) do {} while (0)
#define TR_LEAVE(s) do {} while (0)
#define TR_ROLE(r) do {} while (0)
#endif

// Role assignment.  The matrix pipe is per SIMD, so the four builder waves must sit on four DIFFERENT
// SIMDs; which SIMD a wave lands on is the dispatcher's choice (not a function of the wave index that
// software may rely on), so every wave publishes its SIMD id (HW_REG_HW_ID[5:4]) through LDS and the
// first wave of each SIMD becomes a builder; the other twelve waves get dense gather indices.  (128 VGPRs
// per wave force exactly four waves per SIMD for a 16-wave workgroup; the fallback "lowest remaining
// waves" covers a SIMD that should have none.)
struct WaveRole {
  bool builder;
  int idx;      // builder: 0..3; gather wave: 0..11
};
__device__ __forceinline__ WaveRole assign_roles(char* lds, int wave, int lane) {
  int* tab = reinterpret_cast<int*>(lds);
  const int simd = (int)(__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)) & 3u);   // hwreg(HW_REG_HW_ID, 4, 2)
  if (lane == 0) tab[wave] = simd;
  __syncthreads();
  int seen = 0, builders = 0, bmask = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int sw = uni(tab[w]);
    if (!((seen >> sw) & 1) && builders < NBW) { bmask |= 1 << w; ++builders; }
    seen |= 1 << sw;
  }
#pragma unroll
  for (int w = 0; w < NW; ++w)
    if (builders < NBW && !((bmask >> w) & 1)) { bmask |= 1 << w; ++builders; }
  WaveRole r;
  r.builder = (bmask >> wave) & 1;
  const int below = bmask & ((1 << wave) - 1);
  r.idx = r.builder ? __builtin_popcount(below) : wave - __builtin_popcount(below);
  __syncthreads();   // the table is dead: the first LUT stage may overwrite it
  return r;
}

// ------------------------------------------------------------------------------------------------
// Gather (gather waves).  The wave owns CPW = 2*HC consecutive output channels; lane half h = lane >> 5
// works on channels h*HC .. h*HC+HC-1 and carries images 4*(lane & 31) .. +3 of the panel.  The row
// offsets (uint16, slot * 64 B) of the HC channels of a half are fetched as packed dwords (per-lane
// address: the two halves read different table entries) a stage ahead.  A look-up pair is then
//   v_add_u32_sdwa (lane address + WORD_k of the packed offsets), ds_read_b128, 2 x v_pk_add_f32.
// Blocks of up to eight reads are hand-scheduled: all addresses, all reads back to back, then counted
// s_waitcnt + adds IN PLACE (tied operands: an accumulator never changes register).  The counted waits stay
// correct with other lgkm operations outstanding at entry: LDS returns in order.  `valid` (wave-uniform)
// = 0 skips the block with a branch INSIDE the asm text, so that the compiler sees straight-line code and
// keeps every accumulator in one register for the whole kernel.  The read temporaries are FIXED physical
// registers named in the clobber lists (an asm operand cannot be sliced into the halves the adds need):
// blocks of eight use v[96:127], smaller blocks v[112:127].
// ------------------------------------------------------------------------------------------------
template <int DW>
struct Idx {
  uint32_t w[DW];                                    // pairs of pre-scaled 16-bit row offsets
};
// The plain assignment table holds one-byte row slots, four per dword in the order (e0, e2, e1, e3) (QkSlots): a half-wave's
// entries of one (tap, sub-space) are BW dwords ...
__host__ __device__ constexpr int idx_bytes_dwords(int DW) { return (DW + 1) / 2; }
template <int BW>
struct IdxB {
  uint32_t w[BW];
};
template <int BW>
__device__ __forceinline__ void vload_idx(IdxB<BW>& o, const uint8_t* __restrict__ ap, uint32_t laneOff) {
  const uint32_t* __restrict__ ap4 =
      reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(reinterpret_cast<const char*>(ap) + laneOff, 4));
#pragma unroll
  for (int j = 0; j < BW; ++j) o.w[j] = ap4[j];
}
// ... and become the offset pairs of the look-up blocks with a shift and a mask each
template <int DW>
__device__ __forceinline__ Idx<DW> expand_idx(const IdxB<idx_bytes_dwords(DW)>& b) {
  Idx<DW> o;
#pragma unroll
  for (int j = 0; j < idx_bytes_dwords(DW); ++j) {
    if (2 * j < DW) o.w[2 * j] = (b.w[j] << 6) & 0x1fc01fc0u;
    if (2 * j + 1 < DW) o.w[2 * j + 1] = (b.w[j] >> 2) & 0x1fc01fc0u;
  }
  return o;
}


// eight reads = 16 look-ups; acc[2j], acc[2j+1] = the four images of channel j.  The address of a read lives in the
// first register of its own destination quad (the address is consumed at issue).
__device__ __forceinline__ void gq8(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t base, int valid) {
  asm volatile(Q_SKIP
               Q_AD("v96", "w0", "WORD_0") Q_AD("v100", "w0", "WORD_1") Q_AD("v104", "w1", "WORD_0") Q_AD("v108", "w1", "WORD_1")
               Q_AD("v112", "w2", "WORD_0") Q_AD("v116", "w2", "WORD_1") Q_AD("v120", "w3", "WORD_0") Q_AD("v124", "w3", "WORD_1")
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
               : Q_CLOB8);
}
// four reads = 8 look-ups
__device__ __forceinline__ void gq4(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t base, int valid) {
  asm volatile(Q_SKIP
               Q_AD("v112", "w0", "WORD_0") Q_AD("v116", "w0", "WORD_1") Q_AD("v120", "w1", "WORD_0") Q_AD("v124", "w1", "WORD_1")
               Q_RD("v[112:115]", "v112") Q_RD("v[116:119]", "v116") Q_RD("v[120:123]", "v120") Q_RD("v[124:127]", "v124")
               Q_ACC("3", "c0", "c1", "v[112:113]", "v[114:115]") Q_ACC("2", "c2", "c3", "v[116:117]", "v[118:119]")
               Q_ACC("1", "c4", "c5", "v[120:121]", "v[122:123]") Q_ACC("0", "c6", "c7", "v[124:125]", "v[126:127]")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
                 [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7])
               : [w0] "v"(w0), [w1] "v"(w1), [b] "v"(base), [ok] "s"(valid)
               : Q_CLOB4);
}
// three reads = 6 look-ups; FIRST = 1: entries (w0.lo, w0.hi, w1.lo), FIRST = 0: (w0.hi, w1.lo, w1.hi)
template <int FIRST>
__device__ __forceinline__ void gq3(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t base, int valid) {
  if (FIRST) {
    asm volatile(Q_SKIP
                 Q_AD("v112", "w0", "WORD_0") Q_AD("v116", "w0", "WORD_1") Q_AD("v120", "w1", "WORD_0")
                 Q_RD("v[112:115]", "v112") Q_RD("v[116:119]", "v116") Q_RD("v[120:123]", "v120")
                 Q_ACC("2", "c0", "c1", "v[112:113]", "v[114:115]") Q_ACC("1", "c2", "c3", "v[116:117]", "v[118:119]")
                 Q_ACC("0", "c4", "c5", "v[120:121]", "v[122:123]")
                 "\n.Lqskip%=:"
                 : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
                   [c5] "+v"(acc[5])
                 : [w0] "v"(w0), [w1] "v"(w1), [b] "v"(base), [ok] "s"(valid)
                 : Q_CLOB4);
  } else {
    asm volatile(Q_SKIP
                 Q_AD("v112", "w0", "WORD_1") Q_AD("v116", "w1", "WORD_0") Q_AD("v120", "w1", "WORD_1")
                 Q_RD("v[112:115]", "v112") Q_RD("v[116:119]", "v116") Q_RD("v[120:123]", "v120")
                 Q_ACC("2", "c0", "c1", "v[112:113]", "v[114:115]") Q_ACC("1", "c2", "c3", "v[116:117]", "v[118:119]")
                 Q_ACC("0", "c4", "c5", "v[120:121]", "v[122:123]")
                 "\n.Lqskip%=:"
                 : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
                   [c5] "+v"(acc[5])
                 : [w0] "v"(w0), [w1] "v"(w1), [b] "v"(base), [ok] "s"(valid)
                 : Q_CLOB4);
  }
}
// six reads = 12 look-ups (CPW = 12: all channels of a position in one block instead of two blocks of three — a
// block is a round trip to the LDS, and a wave of a 128-channel layer made up to six of them per stage); temporaries
// v[104:127]
#define Q_CLOB6 "scc", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",   \
                "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
__device__ __forceinline__ void gq6(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t base, int valid) {
  asm volatile(Q_SKIP
               Q_AD("v104", "w0", "WORD_0") Q_AD("v108", "w0", "WORD_1") Q_AD("v112", "w1", "WORD_0") Q_AD("v116", "w1", "WORD_1")
               Q_AD("v120", "w2", "WORD_0") Q_AD("v124", "w2", "WORD_1")
               Q_RD("v[104:107]", "v104") Q_RD("v[108:111]", "v108") Q_RD("v[112:115]", "v112") Q_RD("v[116:119]", "v116")
               Q_RD("v[120:123]", "v120") Q_RD("v[124:127]", "v124")
               Q_ACC("5", "c0", "c1", "v[104:105]", "v[106:107]") Q_ACC("4", "c2", "c3", "v[108:109]", "v[110:111]")
               Q_ACC("3", "c4", "c5", "v[112:113]", "v[114:115]") Q_ACC("2", "c6", "c7", "v[116:117]", "v[118:119]")
               Q_ACC("1", "c8", "c9", "v[120:121]", "v[122:123]") Q_ACC("0", "c10", "c11", "v[124:125]", "v[126:127]")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
                 [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7]), [c8] "+v"(acc[8]), [c9] "+v"(acc[9]),
                 [c10] "+v"(acc[10]), [c11] "+v"(acc[11])
               : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [b] "v"(base), [ok] "s"(valid)
               : Q_CLOB6);
}
// twelve reads = 24 look-ups (CPW = 24: one block instead of eight reads + four); temporaries v[80:127]
__device__ __forceinline__ void gq12(f32x2* acc, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5,
                                     uint32_t base, int valid) {
  asm volatile(Q_SKIP
               Q_AD("v80", "w0", "WORD_0") Q_AD("v84", "w0", "WORD_1") Q_AD("v88", "w1", "WORD_0") Q_AD("v92", "w1", "WORD_1")
               Q_AD("v96", "w2", "WORD_0") Q_AD("v100", "w2", "WORD_1") Q_AD("v104", "w3", "WORD_0") Q_AD("v108", "w3", "WORD_1")
               Q_AD("v112", "w4", "WORD_0") Q_AD("v116", "w4", "WORD_1") Q_AD("v120", "w5", "WORD_0") Q_AD("v124", "w5", "WORD_1")
               Q_RD("v[80:83]", "v80") Q_RD("v[84:87]", "v84") Q_RD("v[88:91]", "v88") Q_RD("v[92:95]", "v92")
               Q_RD("v[96:99]", "v96") Q_RD("v[100:103]", "v100") Q_RD("v[104:107]", "v104") Q_RD("v[108:111]", "v108")
               Q_RD("v[112:115]", "v112") Q_RD("v[116:119]", "v116") Q_RD("v[120:123]", "v120") Q_RD("v[124:127]", "v124")
               Q_ACC("11", "c0", "c1", "v[80:81]", "v[82:83]") Q_ACC("10", "c2", "c3", "v[84:85]", "v[86:87]")
               Q_ACC("9", "c4", "c5", "v[88:89]", "v[90:91]") Q_ACC("8", "c6", "c7", "v[92:93]", "v[94:95]")
               Q_ACC("7", "c8", "c9", "v[96:97]", "v[98:99]") Q_ACC("6", "c10", "c11", "v[100:101]", "v[102:103]")
               Q_ACC("5", "c12", "c13", "v[104:105]", "v[106:107]") Q_ACC("4", "c14", "c15", "v[108:109]", "v[110:111]")
               Q_ACC("3", "c16", "c17", "v[112:113]", "v[114:115]") Q_ACC("2", "c18", "c19", "v[116:117]", "v[118:119]")
               Q_ACC("1", "c20", "c21", "v[120:121]", "v[122:123]") Q_ACC("0", "c22", "c23", "v[124:125]", "v[126:127]")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]), [c5] "+v"(acc[5]), [c6] "+v"(acc[6]), [c7] "+v"(acc[7]), [c8] "+v"(acc[8]), [c9] "+v"(acc[9]), [c10] "+v"(acc[10]), [c11] "+v"(acc[11]), [c12] "+v"(acc[12]), [c13] "+v"(acc[13]), [c14] "+v"(acc[14]), [c15] "+v"(acc[15]), [c16] "+v"(acc[16]), [c17] "+v"(acc[17]), [c18] "+v"(acc[18]), [c19] "+v"(acc[19]), [c20] "+v"(acc[20]), [c21] "+v"(acc[21]), [c22] "+v"(acc[22]), [c23] "+v"(acc[23])
               : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [w4] "v"(w4), [w5] "v"(w5), [b] "v"(base),
                 [ok] "s"(valid)
               : "scc", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}
// two reads = 4 look-ups
__device__ __forceinline__ void gq2(f32x2* acc, uint32_t w0, uint32_t base, int valid) {
  asm volatile(Q_SKIP
               Q_AD("v112", "w0", "WORD_0") Q_AD("v116", "w0", "WORD_1")
               Q_RD("v[112:115]", "v112") Q_RD("v[116:119]", "v116")
               Q_ACC("1", "c0", "c1", "v[112:113]", "v[114:115]") Q_ACC("0", "c2", "c3", "v[116:117]", "v[118:119]")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3])
               : [w0] "v"(w0), [b] "v"(base), [ok] "s"(valid)
               : Q_CLOB4);
}

// dwords of packed offsets per half-wave for CPW channels per wave
__host__ __device__ constexpr int idx_dwords(int CPW) { return (CPW / 2 + 1) / 2; }

// the CPW look-ups (CPW/2 reads) of one position / sub-space; `stage` = LDS byte address of the lane's four
// images in slot 0 of the stage
template <int CPW>
__device__ __forceinline__ void gather_apply(f32x2 (&acc)[CPW], const Idx<idx_dwords(CPW)>& o, uint32_t stage, int validIn) {
  static_assert(CPW == 4 || CPW == 6 || CPW == 8 || CPW == 12 || CPW == 16 || CPW == 24 || CPW == 32, "channel slices");
  const int valid = uni(validIn);       // an "s" operand of the blocks; free when the value already lives in an SGPR
  if constexpr (CPW == 32) {
    gq8(&acc[0], o.w[0], o.w[1], o.w[2], o.w[3], stage, valid);
    gq8(&acc[16], o.w[4], o.w[5], o.w[6], o.w[7], stage, valid);
  } else if constexpr (CPW == 24) {
    gq12(&acc[0], o.w[0], o.w[1], o.w[2], o.w[3], o.w[4], o.w[5], stage, valid);   // (eight + four reads: VGG-16 -0.9 %)
  } else if constexpr (CPW == 16) {
    gq8(&acc[0], o.w[0], o.w[1], o.w[2], o.w[3], stage, valid);
  } else if constexpr (CPW == 12) {
    gq6(&acc[0], o.w[0], o.w[1], o.w[2], stage, valid);   // (two blocks of three reads: conv2 +4 %, conv5 +5 %)
  } else if constexpr (CPW == 8) {
    gq4(&acc[0], o.w[0], o.w[1], stage, valid);
  } else if constexpr (CPW == 6) {
    gq3<1>(&acc[0], o.w[0], o.w[1], stage, valid);
  } else {
    gq2(&acc[0], o.w[0], stage, valid);
  }
}

// ------------------------------------------------------------------------------------------------
// LUT stage builders (builder waves).  A stage covers sub-spaces m0 .. m0+G-1 (those < mEnd), K rows
// each.  The 128-image activation row of dim d of sub-space m starts at
// xbase + xoff0 + (m*Cs + d) * 512 bytes.
// ------------------------------------------------------------------------------------------------

// Where a builder lane finds its activations.  Panels (every layer but possibly the first): dim d of a pixel is a
// 512-byte row of the 128 images.  NCHW (the network input read in place, ConvParams::srcNchw: no pack kernel, one
// HBM round trip less): dim d is a channel plane, an image is C*H*W floats away from the next; lanes of images past
// the end of the batch re-read the last image (their results are never unpacked).
struct XAddr {
  uint32_t dimStride;     // bytes from dim d to dim d+1 of the same pixel and image
  uint32_t mfmaLane[2];   // MFMA builder: byte offset of (image tile it of the wave, image lane & 15, dim lane >> 4)
  uint32_t pairLane[2];   // exact builder: byte offset of images 2*lane and 2*lane+1
};
__device__ __forceinline__ XAddr xaddr_panel(int bw, int lane) {
  XAddr a;
  a.dimStride = XROWB;
  a.mfmaLane[0] = (uint32_t)(lane >> 4) * XROWB + bw * 128 + (lane & 15) * 4;
  a.mfmaLane[1] = a.mfmaLane[0] + 64;
  a.pairLane[0] = lane * 8;
  a.pairLane[1] = lane * 8 + 4;
  return a;
}
// first image of the panel = img0, n images in the batch, C*H*W floats per image, dims (channels) per group Cg <= 4
__device__ __forceinline__ XAddr xaddr_nchw(int bw, int lane, int img0, int n, uint32_t imgBytes, uint32_t planeBytes, int Cg) {
  XAddr a;
  a.dimStride = planeBytes;
  const uint32_t dim = min(lane >> 4, Cg - 1);          // a dim the group does not have is clamped (and zeroed at use)
#pragma unroll
  for (int it = 0; it < 2; ++it)
    a.mfmaLane[it] = dim * planeBytes + (uint32_t)min(img0 + bw * 32 + it * 16 + (lane & 15), n - 1) * imgBytes;
  a.pairLane[0] = (uint32_t)min(img0 + 2 * lane, n - 1) * imgBytes;
  a.pairLane[1] = (uint32_t)min(img0 + 2 * lane + 1, n - 1) * imgBytes;
  return a;
}


// exact: y = ((0 + x0*c0) + x1*c1) + ...  with separately rounded product and sum, the order of the
// reference's saxpy chain (src/CaffeEva.cc:1284-1289, include/BlasWrapper.h:164-184).  Any K <= 128.
// Builder wave bw computes rows bw*ceil(K/4) .. of every sub-space; a lane carries an image pair.
__device__ __forceinline__ void build_stage_exact(char* stage, const char* __restrict__ xbase, uint32_t xoff0, const XAddr& xa,
                                                  const float* __restrict__ ctrd, int K, int Cs, int D, int G, int m0,
                                                  int mEnd, int bw, int lane, int pd = 1) {
  const int kpw = (K + NBW - 1) / NBW;
  const int k0 = bw * kpw;
  const int k1 = min(K, k0 + kpw);
  char* wl = stage + (lane >> 3) * TILEB + (lane & 7) * 8;     // images 2*lane, 2*lane+1: tile lane/8, 8 B in
  const int swz = lane >> 4;                                   // slot swizzle of that tile (tile >> 1)
  for (int g = 0; g < G; ++g) {
    const int m = m0 + g;
    if (m >= mEnd) break;
    const int md = pd > 1 ? m / pd : m;                          // pseudo sub-spaces (K > 128, ConvParams::pd) share their dims
    const int dsel = min(D - md * Cs, Cs);
    const char* __restrict__ xm = xbase + xoff0 + (uint32_t)(md * Cs) * xa.dimStride;
    f32x2 xv[QCNN_MAX_CS];
#pragma unroll
    for (int d = 0; d < QCNN_MAX_CS; ++d) {
      xv[d] = f32x2{0.0f, 0.0f};
      if (d < dsel)
        xv[d] = f32x2{*reinterpret_cast<const float*>(xm + d * xa.dimStride + xa.pairLane[0]),
                      *reinterpret_cast<const float*>(xm + d * xa.dimStride + xa.pairLane[1])};
    }
    const float* __restrict__ cm = ctrd + (size_t)m * Cs * K;
    for (int k = k0; k < k1; ++k) {
      float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
      for (int d = 0; d < QCNN_MAX_CS; ++d) {
        if (d < dsel) {
          const float c = cm[d * K + k];
          v0 = __fadd_rn(v0, __fmul_rn(xv[d].x, c));
          v1 = __fadd_rn(v1, __fmul_rn(xv[d].y, c));
        }
      }
      *reinterpret_cast<f32x2*>(wl + (row_slot(g * K + k) ^ swz) * 64) = f32x2{v0, v1};
    }
  }
}

// MFMA: D[16 rows][16 images] += A[16 rows x 4 dims] * B[4 dims x 16 images] (v_mfma_f32_16x16x4_f32).
// Lane l holds A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)*4 + r][l&15].  A stage is 8 row tiles x 8 image
// tiles; builder wave bw owns image tiles 2*bw, 2*bw+1 and all 8 row tiles.  Row tile i belongs to
// sub-space m0 + i/KT and starts at code word (i % KT)*16  (KT = K/16 in {1, 2, 4, 8}).  Operands are
// fetched into registers (mfma_load) a whole stage period ahead of their use (mfma_store).  All operand
// loads are UNCONDITIONAL, at wave-uniform base + per-lane constant + immediate (device buffers carry
// slack for the over-read of dims / sub-spaces that do not exist); what must not contribute is zeroed
// by a select at use.
template <int KT, int KS>
struct MfmaOps {
  static constexpr int SUBS = 8 / KT;   // sub-spaces per stage
  float a[8][KS];          // code-book operand per row tile and k-step (KS = 1: Cs <= 4 dims, 2: Cs <= 8)
  float b[2][SUBS][KS];    // activation operand per image tile, sub-space and k-step
};

template <int KT, int KS>
__device__ __forceinline__ void mfma_load(MfmaOps<KT, KS>& o, const char* __restrict__ xbase, uint32_t xoff0, const XAddr& xa,
                                          const float* __restrict__ ctrd, int Cs, int m0, int bw, int lane, bool loadA = true) {
  constexpr int K = KT * 16;
  constexpr int SUBS = MfmaOps<KT, KS>::SUBS;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t laneA = lk * K + (li ^ ((uint32_t)bw << 2));           // floats; rows pre-swizzled for the wave's tiles (mfma_pair)
  const float* __restrict__ cbU = ctrd + (size_t)m0 * Cs * K;            // uniform
  const char* __restrict__ xbU = xbase + xoff0 + (uint32_t)(m0 * Cs) * xa.dimStride;   // uniform
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (loadA) {                                             // a layer with one stage per pixel keeps its code book tiles
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int mi = i / KT, kk = (i % KT) * 16;           // compile-time
        o.a[i][ks] = (cbU + ((mi * Cs + ks * 4) * K + kk))[laneA];
      }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub)
        o.b[it][sub][ks] = *reinterpret_cast<const float*>(xbU + (uint32_t)(sub * Cs + ks * 4) * xa.dimStride + xa.mfmaLane[it]);
  }
}

__device__ __forceinline__ f32x4 round_f16(f32x4 v) {
  return f32x4{(float)(_Float16)v[0], (float)(_Float16)v[1], (float)(_Float16)v[2], (float)(_Float16)v[3]};
}

// Multiply the stage `o` was loaded for (m0, mEnd, D, Cs) out into stage buffer BUF.  The 16 tiles of the wave
// are walked in pairs with a hand-made software pipeline: MFMA(pair n) is interleaved instruction by
// instruction with the LDS writes of pair n-1, so that a write (which a single wave can only issue every ~15
// cycles) sits in the shadow of a matrix instruction and never waits for its own result.
// F16: -1 = test the run-time flag after every pair (8-dim layers: removing those uniform branches from the MFMA
// stream made the layers 3 % SLOWER), 0 / 1 = compile-time (4-dim first layers, whose builder chain is the pole of the
// stage: every instruction less counts, -1 %)
template <int KT, int KS, int BUF, int N, int F16 = -1>
__device__ __forceinline__ void mfma_pair(MfmaOps<KT, KS>& o, f32x4& pa, f32x4& pb, int bw, int f16) {
  constexpr int it = (N < 8 ? N : 0) / 4, i = 2 * ((N < 8 ? N : 0) % 4);          // this pair: tiles (it, i), (it, i+1)
  constexpr int pit = (N > 0 ? N - 1 : 0) / 4, pi = 2 * ((N > 0 ? N - 1 : 0) % 4);  // previous pair (results in pa, pb)
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  const uint32_t m0v = (uint32_t)BUF * STAGE_BYTES + (uint32_t)(2 * bw + pit) * TILEB;
  f32x4 ca = zero, cb = zero;
  if constexpr (KS == 1) {        // one k-step: MFMA, the four stores of one tile, MFMA, the four stores of the other
    __builtin_amdgcn_sched_barrier(0);
    if (N < 8) ca = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][0], o.b[it][i / KT][0], zero, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (N > 0) store_tile_all<pi>(pa, m0v);
    __builtin_amdgcn_sched_barrier(0);
    if (N < 8) cb = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i + 1][0], o.b[it][(i + 1) / KT][0], zero, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (N > 0) store_tile_all<pi + 1>(pb, m0v);
    __builtin_amdgcn_sched_barrier(0);
    pa = ca; pb = cb;
    if (F16 < 0 ? f16 != 0 : F16 == 1) { pa = round_f16(pa); pb = round_f16(pb); }
    return;
  }
  __builtin_amdgcn_sched_barrier(0);
  if (N < 8) ca = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][0], o.b[it][i / KT][0], zero, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if (N > 0) store_tile_lo<pi>(pa, m0v);
  __builtin_amdgcn_sched_barrier(0);
  if (N < 8) cb = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i + 1][0], o.b[it][(i + 1) / KT][0], zero, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if (N > 0) store_tile_hi<pi>(pa, m0v);
  __builtin_amdgcn_sched_barrier(0);
  if (KS > 1 && N < 8) ca = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][KS - 1], o.b[it][i / KT][KS - 1], ca, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if (N > 0) store_tile_lo<pi + 1>(pb, m0v);
  __builtin_amdgcn_sched_barrier(0);
  if (KS > 1 && N < 8)
    cb = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i + 1][KS - 1], o.b[it][(i + 1) / KT][KS - 1], cb, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  if (N > 0) store_tile_hi<pi + 1>(pb, m0v);
  __builtin_amdgcn_sched_barrier(0);
  pa = ca; pb = cb;
  if (F16 < 0 ? f16 != 0 : F16 == 1) { pa = round_f16(pa); pb = round_f16(pb); }   // tolerance study only
}

template <int KT, int KS, int BUF, int F16>
__device__ __forceinline__ void mfma_multiply(MfmaOps<KT, KS>& o, int bw, int f16) {
  f32x4 pa = {0.0f, 0.0f, 0.0f, 0.0f}, pb = pa;   // results of the previous pair, still to be written
  mfma_pair<KT, KS, BUF, 0, F16>(o, pa, pb, bw, f16);
  mfma_pair<KT, KS, BUF, 1, F16>(o, pa, pb, bw, f16);
  mfma_pair<KT, KS, BUF, 2, F16>(o, pa, pb, bw, f16);
  mfma_pair<KT, KS, BUF, 3, F16>(o, pa, pb, bw, f16);
  mfma_pair<KT, KS, BUF, 4, F16>(o, pa, pb, bw, f16);
  mfma_pair<KT, KS, BUF, 5, F16>(o, pa, pb, bw, f16);
  mfma_pair<KT, KS, BUF, 6, F16>(o, pa, pb, bw, f16);
  mfma_pair<KT, KS, BUF, 7, F16>(o, pa, pb, bw, f16);
  mfma_pair<KT, KS, BUF, 8, F16>(o, pa, pb, bw, f16);
}

// zero what must not contribute: dims a sub-space does not have, sub-spaces past the end.  maskA = false leaves the
// code-book operands alone (they were masked when they were loaded and are kept in registers: mfma_load's loadA)
template <int KT, int KS>
__device__ __forceinline__ void mfma_mask(MfmaOps<KT, KS>& o, int Cs, int D, int m0, int mEnd, int lane, bool maskA, bool maskB) {
  constexpr int SUBS = MfmaOps<KT, KS>::SUBS;
  const int lk = lane >> 4;
  // plain: every sub-space of the stage exists and has all 4*KS dims -> nothing to zero
  const bool plain = (m0 + SUBS <= mEnd) && (D - (m0 + SUBS - 1) * Cs >= 4 * KS) && (Cs == 4 * KS);
  if (!plain) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        const bool ok = (m0 + sub < mEnd) && (ks * 4 + lk < min(D - (m0 + sub) * Cs, Cs));
        if (maskB) {
#pragma unroll
          for (int it = 0; it < 2; ++it) o.b[it][sub][ks] = ok ? o.b[it][sub][ks] : 0.0f;
        }
        if (maskA) {
#pragma unroll
          for (int i = sub * KT; i < (sub + 1) * KT; ++i) o.a[i][ks] = ok ? o.a[i][ks] : 0.0f;
        }
      }
    }
  }
}

template <int KT, int KS, int BUF>
__device__ __forceinline__ void mfma_store(MfmaOps<KT, KS>& o, int Cs, int D, int m0, int mEnd, int bw, int lane, int f16,
                                           bool maskA = true) {
  if (maskA) {
    mfma_mask<KT, KS>(o, Cs, D, m0, mEnd, lane, true, true);
  } else {                        // kept operands: one sub-space, always the same -> the activation mask is lane-constant
    const bool ok = (lane >> 4) < min(D, Cs);
#pragma unroll
    for (int it = 0; it < 2; ++it) o.b[it][0][0] = ok ? o.b[it][0][0] : 0.0f;
  }
  if constexpr (KS == 1) {
    if (f16) mfma_multiply<KT, KS, BUF, 1>(o, bw, f16); else mfma_multiply<KT, KS, BUF, 0>(o, bw, f16);
  } else {
    mfma_multiply<KT, KS, BUF, -1>(o, bw, f16);
  }
}


// ------------------------------------------------------------------------------------------------
// conv.  Workgroup = 16 waves = one 128-image panel x a TH x TW tile of output positions x 12*CPW output
// channels of one group.  Stages (source pixel row-major, sub-space group ascending) are double buffered in
// LDS, one s_barrier per stage.  The waves are SPECIALISED, so that the chain "MFMA -> LDS write" of stage
// s+1 and the chain "LDS read -> add" of stage s run concurrently instead of one after the other in every
// wave:
//   builder waves (one per SIMD):  [build stage s+1 from operands in registers] [fetch operands s+2]
//   gather waves:                  [prefetch offsets of stage s+1] [gather stage s]
// Gather wave gw owns channels gw*CPW .. +CPW-1 of the workgroup's slice for ALL TH*TW positions and keeps
// TH*TW x CPW float2 accumulators.  KT = K/16 selects the MFMA builder, KT = 0 the exact builder (any
// K <= 128).
// ------------------------------------------------------------------------------------------------
// offsets of the first sub-space of stage c for every position of the tile (taps that do not exist are
// clamped to an existing one: the load is harmless, the gather skips them)
template <int TH, int TW, int CPW>
__device__ __forceinline__ void conv_prefetch_idx(IdxB<idx_bytes_dwords(idx_dwords(CPW))> (&v)[TH * TW], const StagePos& c,
                                                  const ConvGeom& g, const uint8_t* __restrict__ rowsW, const int (&rowStart)[TH],
                                                  const int (&colStart)[TW], uint32_t laneOff) {
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) {
    const int kh = min(max(c.hi - rowStart[dy], 0), g.knl - 1);
#pragma unroll
    for (int dx = 0; dx < TW; ++dx) {
      const int kw = min(max(c.wi - colStart[dx], 0), g.knl - 1);
      vload_idx(v[dy * TW + dx], rowsW + (size_t)((kh * g.knl + kw) * g.M + c.mg * g.G) * g.rowStride, laneOff);
    }
  }
}

template <int TH, int TW, int CPW, bool ONE>
__device__ __forceinline__ void conv_gather(f32x2 (&acc)[TH * TW][CPW],
                                            const IdxB<idx_bytes_dwords(idx_dwords(CPW))> (&first)[TH * TW],
                                            const StagePos& c, const ConvGeom& g, const uint8_t* __restrict__ rowsW,
                                            const int (&rowStart)[TH], const int (&colStart)[TW], uint32_t stage,
                                            uint32_t laneOff, bool live) {
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) {
    const int kh = c.hi - rowStart[dy];
    const bool rowOk = live && (unsigned)kh < (unsigned)g.knl;
#pragma unroll
    for (int dx = 0; dx < TW; ++dx) {
      const int kw = c.wi - colStart[dx];
      const int valid = uni((rowOk && (unsigned)kw < (unsigned)g.knl) ? 1 : 0);
      gather_apply<CPW>(acc[dy * TW + dx], expand_idx<idx_dwords(CPW)>(first[dy * TW + dx]), stage, valid);
      if (!ONE) {                              // further sub-spaces of the stage (K <= 64 only)
        const int m0 = c.mg * g.G;
        const int n = valid ? min(g.M, m0 + g.G) - m0 : 0;
        for (int i = 1; i < n; ++i) {
          IdxB<idx_bytes_dwords(idx_dwords(CPW))> more;
          vload_idx(more, rowsW + (size_t)((kh * g.knl + kw) * g.M + m0 + i) * g.rowStride, laneOff);
          gather_apply<CPW>(acc[dy * TW + dx], expand_idx<idx_dwords(CPW)>(more), stage, 1);
        }
      }
    }
  }
}


// ---- offsets through the program table (QkProgram, qcnn_kernels.h): K = 128 panel kernels ----
// One wave fetches the workgroup's contiguous row of the stage after next straight into LDS (LDS-DMA: no registers,
// no ds_write); every gather wave then picks its block for the NEXT stage with one or two ds_read_b128 while it
// gathers the current one.  Measured before the change: the per-position table reads and their address arithmetic
// kept a gather wave busy for 565-1270 cycles per stage (profiles/r2_v7/trace); dropping them altogether (wrong
// results, timing only) was worth 11 % of the whole forward pass.

template <int TH, int TW, int CPW, int NB>
__device__ __forceinline__ void conv_gather_prog(f32x2 (&acc)[TH * TW][CPW], const IdxBlk<NB>& blk, const StagePos& c,
                                                 const ConvGeom& g, const int (&rowStart)[TH], const int (&colStart)[TW],
                                                 uint32_t stage, int live) {
  constexpr int DW = idx_dwords(CPW);
  int colOk[TW];
#pragma unroll
  for (int dx = 0; dx < TW; ++dx) colOk[dx] = in_range(c.wi - colStart[dx], g.knl);
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) {
    const int rowOk = live & in_range(c.hi - rowStart[dy], g.knl);
#pragma unroll
    for (int dx = 0; dx < TW; ++dx) {
      Idx<DW> o;
#pragma unroll
      for (int j = 0; j < DW; ++j) o.w[j] = blk.w[(dy * TW + dx) * DW + j];
      gather_apply<CPW>(acc[dy * TW + dx], o, stage, rowOk & colOk[dx]);
    }
  }
}

// sliding variant: slot q currently holds an output row whose window starts at source row xq[q] (okq[q] = 0 when that
// row lies outside the segment); it looks at source row c.hi through tap row d = c.hi - xq[q], 0 <= d < knl.  A strip of
// NC output columns keeps NS slots per column (accumulators [column][slot]); column dx looks at source column c.wi
// through tap column c.wi - colStart[dx]
template <int NC, int NS, int CPW, int NB>
__device__ __forceinline__ void conv_gather_slide(f32x2 (&acc)[NC * NS][CPW], const IdxBlk<NB>& blk, const StagePos& c, int knl,
                                                  const int (&xq)[NS], const int (&okq)[NS], const int (&colStart)[NC],
                                                  uint32_t stage, int live) {
  constexpr int DW = idx_dwords(CPW);
#pragma unroll
  for (int dx = 0; dx < NC; ++dx) {
    const int colOk = NC == 1 ? live : live & in_range(c.wi - colStart[dx], knl);
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      Idx<DW> o;
#pragma unroll
      for (int j = 0; j < DW; ++j) o.w[j] = blk.w[(dx * NS + q) * DW + j];
      gather_apply<CPW>(acc[dx * NS + q], o, stage, colOk & okq[q] & in_range(c.hi - xq[q], knl));
    }
  }
}

// SLIDE (K = 128, MFMA builders): instead of a fixed TH x TW tile the workgroup owns a SEGMENT [segBeg, segEnd) of one
// output COLUMN and sweeps the source rows under it from top to bottom (stages in the usual row-major order: the pixels
// of a source row are neighbours in memory, so a first layer that reads the NCHW input in place keeps its cache lines).
// A source row is looked at by ceil(knl / stride) output rows only, so TW = that many accumulator SLOTS suffice: slot q
// holds output row segBeg + q, then + TW, ... — when the last row of a position's window has been consumed its sums are
// stored and the slot starts over with the bias for the position TW rows further down.  Every source pixel of the strip
// is built ONCE per segment (the tile kernel rebuilds it for every tile whose receptive field holds it): conv1 of AlexNet
// 56 -> 44 stages per output position, a 3x3 / 1 layer with 128 channels 5 -> 3.5.  (The template's TH = 1, TW = slots.)
template <int TH, int TW, int CPW, int KT, int KS, bool SLIDE = false>
__global__ __launch_bounds__(NW * 64) void k_conv_aprx(ConvParams p, int tilesX, int tilesY, int chunksPerGrp, int G,
                                                        int rowStride) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  static_assert(!SLIDE || (KT == 8 && TH <= 2), "the sliding variant exists for the program-table kernels; TH = output columns of the strip");
  constexpr int NP = TH * TW;
  constexpr int HC = CPW / 2;
  constexpr int DW = idx_dwords(CPW);
  constexpr int KTT = KT > 0 ? KT : 1;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  // blockIdx.x = tile rank (heaviest first) * panels + panel for the tiles a single workgroup runs from end to end; the
  // ranks from p.splitFrom on (the tail of a launch that does not fill the chip: ConvParams::splitZ) follow, each cut into
  // splitZ workgroups that take consecutive slices of the tile's stage sequence and write partial sums
  int ty = 0, tx = 0, panel, slice = 0, slices = 1, tailTile = 0;
  int segBeg = 0, segEnd = 0;                       // SLIDE: this workgroup's output rows of column tx
  if constexpr (SLIDE) {
    // blockIdx.x = (segment-major unit, longest segments first) * panels + panel
    const unsigned unit = blockIdx.x / (unsigned)p.panels;
    panel = (int)(blockIdx.x % (unsigned)p.panels);
    const unsigned colGroups = (unsigned)(p.Wo + TH - 1) / TH;    // SLIDE: TH = output columns of the strip (1 or 2)
    const int seg = (int)(unit / colGroups);
    tx = (int)(unit % colGroups) * TH;                 // the strip's first output column
    segBeg = p.segBeg[seg]; segEnd = p.segBeg[seg + 1];   // its output rows
  } else {
    const unsigned bx = blockIdx.x, nBody = (unsigned)p.splitFrom * (unsigned)p.panels;
    int rank;
    if (bx < nBody) {
      rank = (int)(bx / (unsigned)p.panels);
      panel = (int)(bx % (unsigned)p.panels);
    } else {
      const unsigned r = bx - nBody;
      slices = p.splitZ;
      panel = (int)(r % (unsigned)p.panels);
      slice = (int)((r / (unsigned)p.panels) % (unsigned)slices);
      tailTile = (int)(r / (unsigned)(p.panels * slices));
      rank = p.splitFrom + tailTile;
    }
    tile_of_rank(rank, tilesY, tilesX, ty, tx);
  }
  const int grp = blockIdx.y / chunksPerGrp, chunk = blockIdx.y % chunksPerGrp;
  const int Cg = p.Cin / p.grp, Ctg = p.Ct / p.grp;
  const int M = p.M;

  const int ho0 = SLIDE ? segBeg : ty * TH, wo0 = SLIDE ? tx : tx * TW;
  const int hoL = SLIDE ? segEnd - 1 : min(ho0 + TH, p.Ho) - 1;                                   // last real position
  const int woL = SLIDE ? min(tx + TH, p.Wo) - 1 : min(wo0 + TW, p.Wo) - 1;
  const int hiL = max(0, ho0 * p.stride - p.pad), hiU = min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1);
  ConvGeom g;
  g.W = p.W; g.Cin = p.Cin; g.knl = p.knl; g.M = M; g.G = G; g.rowStride = (uint32_t)rowStride;
  g.pixStride = p.srcNchw ? 4u : (uint32_t)p.Cin * (uint32_t)XROWB;
  g.MG = (M + G - 1) / G;                           // stages per source pixel
  g.wiL = max(0, wo0 * p.stride - p.pad);
  g.wiU = min(p.W - 1, woL * p.stride - p.pad + p.knl - 1);
  g.slide = SLIDE ? 1 : 0; g.hiL = hiL; g.hiU = hiU; g.period = TW * p.stride;
  const int cols = g.wiU - g.wiL + 1;
  const int Stot = (hiU - hiL + 1) * cols * g.MG;   // stages of the whole tile; this workgroup runs [sBeg, sBeg + S)
  const int sBeg = (int)((long long)Stot * slice / slices);
  const int S = (int)((long long)Stot * (slice + 1) / slices) - sBeg;
  const int Sp = (S + 1) & ~1;                      // every wave runs Sp stage periods (barriers)
  const StagePos first = {hiL + (sBeg / g.MG) / cols, g.wiL + (sBeg / g.MG) % cols, sBeg % g.MG,
                          SLIDE ? (int)((unsigned)(hiL - (ho0 * p.stride - p.pad)) % (unsigned)(TW * p.stride)) : 0};
  if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();   // the stage addressing assumes the dynamic segment starts at LDS byte 0

  const WaveRole role = assign_roles(lds, wave, lane);
  TR_ROLE(role);
  if (role.builder) {
    // ---------------------------------------------------------------- builder wave ----
    const int bw = uni(role.idx);                   // kept in an SGPR: it feeds the M0 operand of the add-TID stores
    const int K = p.K, Cs = p.Cs;
    __builtin_amdgcn_s_setprio(QCNN_PRIO_BUILDER);
    // activations: this panel's rows — or, for a first layer reading the NCHW network input in place, the batch itself
    const char* __restrict__ xbase = p.srcNchw
        ? reinterpret_cast<const char*>(p.src + (size_t)grp * Cg * p.H * p.W)
        : reinterpret_cast<const char*>(p.src + ((size_t)panel * p.H * p.W * p.Cin + (size_t)grp * Cg) * PANEL);
    const XAddr xa = p.srcNchw ? xaddr_nchw(bw, lane, (p.panel0 + panel) * PANEL, p.nImages, (uint32_t)p.Cin * p.H * p.W * 4u,
                                            (uint32_t)p.H * p.W * 4u, Cg)
                               : xaddr_panel(bw, lane);
    {
    // Two operand sets: while stage s+1 is multiplied out of one, the other one already holds (or is
    // receiving) stage s+2, and the loads of stage s+3 are issued as soon as the first is consumed, so
    // an operand fetch has a whole stage period to land.
    MfmaOps<KTT, KS> opsA, opsB;
    // One stage per source pixel (a first layer: M = 1, <= 4 input channels): the code book operands never change and
    // stay in registers (conv1 -3 %).  Only the KS = 1 instantiations carry the test: as a run-time branch in every
    // kernel it cost the 8-dim layers 2-5 % (the operand loads are no longer straight-line code).
    const bool reloadA = KS != 1 || g.MG > 1;
    StagePos q1 = next_pos(first, g);
    StagePos q2 = next_pos(q1, g);
    StagePos q3 = next_pos(q2, g);
    if (KT > 0) {
      mfma_load<KTT, KS>(opsA, xbase, pixel_off(first, g), xa, p.ctrd, Cs, first.mg * G, bw, lane);
      mfma_store<KTT, KS, 0>(opsA, Cs, Cg, first.mg * G, M, bw, lane, p.lutF16);
      {
        const StagePos qa = (1 < S) ? q1 : first, qb = (2 < S) ? q2 : first;   // stages past the end re-fetch the first
        mfma_load<KTT, KS>(opsA, xbase, pixel_off(qa, g), xa, p.ctrd, Cs, qa.mg * G, bw, lane);
        __builtin_amdgcn_sched_barrier(0);             // keep set A's loads older than set B's (vmcnt accounting)
        mfma_load<KTT, KS>(opsB, xbase, pixel_off(qb, g), xa, p.ctrd, Cs, qb.mg * G, bw, lane);
        if (!reloadA) {                                  // kept code-book operands are masked here, once
          mfma_mask<KTT, KS>(opsA, Cs, Cg, 0, M, lane, true, false);
          mfma_mask<KTT, KS>(opsB, Cs, Cg, 0, M, lane, true, false);
        }
      }
    } else {
      build_stage_exact(lds, xbase, pixel_off(first, g), xa, p.ctrd, K, Cs, Cg, G, first.mg * G, M, bw, lane, p.pd);
    }
    barrier_after_lds_writes();
    // Straight-line body (no VMEM operation under a condition), so that the compiler's vmcnt waits are
    // exact: "all but the loads of the other set".  Sp rounds S up to even; the surplus stage is built
    // from re-fetched operands into a buffer nobody reads.
    for (int s = 0; s < Sp; s += 2) {
      if (KT > 0) {                                    // stage s+1 -> buffer 1, from set A
        mfma_store<KTT, KS, 1>(opsA, Cs, Cg, reloadA ? q1.mg * G : 0, M, bw, lane, p.lutF16, reloadA);
        TR_MID(s);
        const StagePos qf = (s + 3 < S) ? q3 : first;
        mfma_load<KTT, KS>(opsA, xbase, pixel_off(qf, g), xa, p.ctrd, Cs, qf.mg * G, bw, lane, reloadA);
      } else if (s + 1 < S) {
        build_stage_exact(lds + STAGE_BYTES, xbase, pixel_off(q1, g), xa, p.ctrd, K, Cs, Cg, G, q1.mg * G, M, bw, lane, p.pd);
      }
      TR_ARRIVE(s);
      barrier_after_lds_writes();
      TR_LEAVE(s);
      q1 = q2; q2 = q3; q3 = next_pos(q3, g);
      if (KT > 0) {                                    // stage s+2 -> buffer 0, from set B
        mfma_store<KTT, KS, 0>(opsB, Cs, Cg, reloadA ? q1.mg * G : 0, M, bw, lane, p.lutF16, reloadA);
        TR_MID(s + 1);
        const StagePos qf = (s + 4 < S) ? q3 : first;
        mfma_load<KTT, KS>(opsB, xbase, pixel_off(qf, g), xa, p.ctrd, Cs, qf.mg * G, bw, lane, reloadA);
      } else if (s + 2 < S) {
        build_stage_exact(lds, xbase, pixel_off(q1, g), xa, p.ctrd, K, Cs, Cg, G, q1.mg * G, M, bw, lane, p.pd);
      }
      TR_ARRIVE(s + 1);
      barrier_after_lds_writes();
      TR_LEAVE(s + 1);
      q1 = q2; q2 = q3; q3 = next_pos(q3, g);
    }
    return;
    }
  }

  // ------------------------------------------------------------------ gather wave ----
  const int gw = role.idx;
  const int half = lane >> 5, quad = lane & 31;
  const int cw0 = (chunk * NGW + gw) * CPW;          // first channel of this wave inside the group
  const bool active = cw0 < Ctg;                     // waves without channels only keep the barriers
  const int cl0 = cw0 + half * HC;                   // first channel of this lane's half inside the group
  // this wave's entries inside one (tap, sub-space) row of the offset table; the two halves read different ones
  constexpr int BW = idx_bytes_dwords(DW);
  const uint8_t* __restrict__ rowsW = p.rows + (size_t)((grp * chunksPerGrp + chunk) * NGW + gw) * 2 * (4 * BW);
  const uint32_t laneOff = (uint32_t)half * (4 * BW);       // bytes
  // XOR-ed with a pre-scaled row offset it gives the lane's read address: tile, slot swizzle (tile >> 1), 16-byte quarter
  const uint32_t laneLds = (uint32_t)(quad >> 2) * TILEB | (uint32_t)(quad >> 3) * 64 | (uint32_t)(quad & 3) * 16;

  f32x2 acc[NP][CPW];
  {
    const float* __restrict__ bp = p.bias + grp * Ctg + (active ? cl0 : 0);   // reads past the last channel stay inside the arena
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const float b = (slice == 0) ? bp[j] : 0.0f;       // the bias enters the first slice's partial sum only
#pragma unroll
      for (int q = 0; q < NP; ++q) { acc[q][2 * j] = f32x2{b, b}; acc[q][2 * j + 1] = f32x2{b, b}; }
    }
  }
  // first source row / column of every position; positions outside the map get a start that can never match a tap
  int rowStart[TH], colStart[TW];
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) rowStart[dy] = (ho0 + dy < p.Ho) ? (ho0 + dy) * p.stride - p.pad : -(1 << 28);
#pragma unroll
  for (int dx = 0; dx < TW; ++dx) colStart[dx] = (wo0 + dx < p.Wo) ? (wo0 + dx) * p.stride - p.pad : -(1 << 28);

  __builtin_amdgcn_s_setprio(QCNN_PRIO_GATHER);
  if constexpr (KT == 8) {
    // Offsets through the program table.  Per stage: [pick the block of stage s+1 from LDS][wave 0: DMA the row of
    // stage s+2 into the other row buffer][gather stage s][barrier].
    constexpr int NB = (NP * DW + 3) / 4 * 4;          // dwords of one wave half's block
    constexpr int WGROW = NGW * 2 * NB * 4;            // bytes of the workgroup's row of one program entry
    // program entry of a stage: (row, column) of its pixel relative to the unclipped receptive field of the tile — or,
    // sliding, (source row modulo TW * stride: the slot -> tap-row map repeats with that period, tap column)
    const int rfW = SLIDE ? (TH - 1) * p.stride + p.knl : (TW - 1) * p.stride + p.knl;
    const int ry0 = ho0 * p.stride - p.pad, rx0 = wo0 * p.stride - p.pad;   // origin of the unclipped receptive field
    const uint32_t entryB = (uint32_t)(p.grp * chunksPerGrp) * WGROW;
    const char* __restrict__ progWg =
        reinterpret_cast<const char*>(SLIDE ? p.progS : p.prog) + (size_t)(grp * chunksPerGrp + chunk) * WGROW;
    auto rowOf = [&](const StagePos& q, int idx) {     // stages past the end: any existing row
      const StagePos c = (idx < S) ? q : first;
      const int row = SLIDE ? c.ph : c.hi - ry0;
      return progWg + (size_t)(uint32_t)((row * rfW + (c.wi - rx0)) * M + c.mg) * entryB;
    };
    // sliding: output column of every slot, the bias pointer for a slot's restart, the store of a finished position
    int colS[TH];                                       // sliding: first source column of every strip column's window
#pragma unroll
    for (int dx = 0; dx < TH; ++dx) colS[dx] = (wo0 + dx < p.Wo) ? (wo0 + dx) * p.stride - p.pad : -(1 << 28);
    int woq[TW], xq[TW], okq[TW];                       // slot state: output row, first source row of its window, in-segment
#pragma unroll
    for (int q = 0; q < TW; ++q) {
      woq[q] = segBeg + q;
      xq[q] = woq[q] * p.stride - p.pad;
      okq[q] = in_range(q, segEnd - segBeg);
    }
    // a slot restarts from the bias every few stages: waves with few channels keep their bias values in registers, the
    // others (12 channels and more: no register to spare, but also many stages per source column) re-read them
    constexpr bool BIAS_REGS = SLIDE && CPW <= 8 && NP * CPW <= 24;
    // wave-uniform bases (SGPR pairs) + one 32-bit lane offset each: no per-lane 64-bit pointers in the register file
    const float* __restrict__ biasU = p.bias + grp * Ctg + (active ? cw0 : 0);
    const uint32_t biasLane = (uint32_t)(half * HC);
    float biasR[BIAS_REGS ? HC : 1];
    if constexpr (BIAS_REGS) {
#pragma unroll
      for (int j = 0; j < HC; ++j) biasR[j] = biasU[biasLane + j];
    }
    float* __restrict__ dstColU = p.dst + ((size_t)panel * p.Ho * p.Wo + (size_t)wo0) * p.Ct * PANEL +
                                  (size_t)(grp * Ctg + (active ? cw0 : 0)) * PANEL;
    const uint32_t dstLane = (uint32_t)(half * HC) * PANEL + 4u * (uint32_t)quad;
    // after the last stage of a source row: positions whose window ends with this row (or with the strip) are stored and
    // their slot restarts from the bias for the position TW rows further down
    auto column_end = [&](const StagePos& c, int live) {
      if (!(live && c.wi == g.wiU && c.mg == g.MG - 1)) return;
#pragma unroll
      for (int q = 0; q < TW; ++q) {
        if ((c.hi - xq[q] == p.knl - 1 || c.hi == hiU) && okq[q]) {
#pragma unroll
          for (int dx = 0; dx < TH; ++dx) {
            float* o = dstColU + ((size_t)woq[q] * p.Wo + dx) * p.Ct * PANEL;      // uniform
            const bool colReal = wo0 + dx < p.Wo;
#pragma unroll
            for (int j = 0; j < HC; ++j) {
              if (colReal && cl0 + j < Ctg) {
                f32x4 v = {acc[dx * TW + q][2 * j].x, acc[dx * TW + q][2 * j].y, acc[dx * TW + q][2 * j + 1].x,
                           acc[dx * TW + q][2 * j + 1].y};
                if (p.relu) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = (0.0f < v[e]) ? v[e] : 0.0f;
                }
                *reinterpret_cast<f32x4*>(o + dstLane + j * PANEL) = v;
              }
              const float b = BIAS_REGS ? biasR[BIAS_REGS ? j : 0] : biasU[biasLane + j];
              acc[dx * TW + q][2 * j] = f32x2{b, b}; acc[dx * TW + q][2 * j + 1] = f32x2{b, b};
            }
          }
          woq[q] += TW;
          xq[q] += TW * p.stride;
          okq[q] = in_range(woq[q] - segBeg, segEnd - segBeg);
        }
      }
    };
    const uint32_t myBlk = (uint32_t)(gw * 2 + half) * NB * 4;
    const bool loader = gw == 0;
    IdxBlk<NB> ba, bb;
    const int activeI = in_range(cw0, Ctg);            // `active` as an integer (see in_range)
    StagePos c0p = first;
    StagePos c1p = next_pos(c0p, g);
    StagePos c2p = next_pos(c1p, g);
    StagePos cEnd = first;                             // sliding: the stage gathered last (its column may have ended)
    int liveEnd = 0;
    blk_load(ba, rowOf(c0p, 0) + myBlk);
    if (loader) idx_row_to_lds<WGROW>(rowOf(c1p, 1), IDX_LDS + IDX_BUF, lane);
    barrier_after_lds_dma();
    for (int s = 0; s < Sp; s += 2) {
      blk_load(bb, lds + IDX_LDS + IDX_BUF + myBlk);                      // stage s+1
      if (loader) idx_row_to_lds<WGROW>(rowOf(c2p, s + 2), IDX_LDS, lane);       // stage s+2
      TR_MID(s);
      if constexpr (SLIDE) {
        // the positions the PREVIOUS stage finished are stored first: their stores then have this whole stage period
        // before the barrier's vmcnt(0) (which the row DMA needs) would wait for them
        column_end(cEnd, liveEnd);
        conv_gather_slide<TH, TW, CPW, NB>(acc, ba, c0p, p.knl, xq, okq, colS, laneLds, activeI);
        cEnd = c0p; liveEnd = activeI;
      } else {
        conv_gather_prog<TH, TW, CPW, NB>(acc, ba, c0p, g, rowStart, colStart, laneLds, activeI);   // S >= 1: stage s exists
      }
      c0p = c1p; c1p = c2p; c2p = next_pos(c2p, g);
      TR_ARRIVE(s);
      barrier_after_lds_dma();
      TR_LEAVE(s);
      blk_load(ba, lds + IDX_LDS + myBlk);                                // stage s+2
      if (loader) idx_row_to_lds<WGROW>(rowOf(c2p, s + 3), IDX_LDS + IDX_BUF, lane);   // stage s+3
      TR_MID(s + 1);
      if constexpr (SLIDE) {
        column_end(cEnd, liveEnd);
        conv_gather_slide<TH, TW, CPW, NB>(acc, bb, c0p, p.knl, xq, okq, colS, laneLds | STAGE_BYTES, activeI & in_range(s + 1, S));
        cEnd = c0p; liveEnd = activeI & in_range(s + 1, S);
      } else {
        conv_gather_prog<TH, TW, CPW, NB>(acc, bb, c0p, g, rowStart, colStart, laneLds | STAGE_BYTES,
                                          activeI & in_range(s + 1, S));
      }
      c0p = c1p; c1p = c2p; c2p = next_pos(c2p, g);
      TR_ARRIVE(s + 1);
      barrier_after_lds_dma();
      TR_LEAVE(s + 1);
    }
    if constexpr (SLIDE) column_end(cEnd, liveEnd);    // the strip's last column
  } else {
    // Offsets from the plain table.  Per stage: [prefetch the offsets of stage s+1][gather stage s][barrier]; two
    // offset sets alternate.
    IdxB<BW> ia[NP], ib[NP];
    StagePos c0p = first;
    StagePos c1p = next_pos(c0p, g);
    conv_prefetch_idx<TH, TW, CPW>(ia, c0p, g, rowsW, rowStart, colStart, laneOff);
    barrier_plain();
    for (int s = 0; s < Sp; s += 2) {
      conv_prefetch_idx<TH, TW, CPW>(ib, c1p, g, rowsW, rowStart, colStart, laneOff);    // stage s+1 (past the end: clamped, unused)
      TR_MID(s);
      conv_gather<TH, TW, CPW, false>(acc, ia, c0p, g, rowsW, rowStart, colStart, laneLds, laneOff, active);
      c0p = c1p; c1p = next_pos(c1p, g);
      TR_ARRIVE(s);
      barrier_plain();
      TR_LEAVE(s);
      conv_prefetch_idx<TH, TW, CPW>(ia, c1p, g, rowsW, rowStart, colStart, laneOff);    // stage s+2
      TR_MID(s + 1);
      conv_gather<TH, TW, CPW, false>(acc, ib, c0p, g, rowsW, rowStart, colStart, laneLds | STAGE_BYTES, laneOff,
                                      active && s + 1 < S);
      c0p = c1p; c1p = next_pos(c1p, g);
      TR_ARRIVE(s + 1);
      barrier_plain();
      TR_LEAVE(s + 1);
    }
  }

  if (active && !SLIDE) {          // (sliding: every position was stored when its window closed)
    // final map, or — a slice of a split tile — this slice's slab of partial sums [tail tile][slice][panel][position]
    // [Ct][128], which k_conv_sum adds up in slice order (ReLU there)
    const bool part = slices > 1;
    float* __restrict__ dst = part ? p.partial + ((size_t)(tailTile * slices + slice) * p.panels + panel) * NP * p.Ct * PANEL
                                   : p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int ho = ho0 + q / TW, wo = wo0 + q % TW;
      if (ho < p.Ho && wo < p.Wo) {
        float* o = dst + ((size_t)(part ? q : ho * p.Wo + wo) * p.Ct + grp * Ctg + cl0) * PANEL + 4 * quad;
#pragma unroll
        for (int j = 0; j < HC; ++j) {
          if (cl0 + j < Ctg) {
            f32x4 v = {acc[q][2 * j].x, acc[q][2 * j].y, acc[q][2 * j + 1].x, acc[q][2 * j + 1].y};
            if (p.relu && !part) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (0.0f < v[e]) ? v[e] : 0.0f;
            }
            *reinterpret_cast<f32x4*>(o + j * PANEL) = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// SYMMETRIC workgroup for layers with exactly 128 channels per group (AlexNet conv2): all 16 waves build AND gather.
// With four builder waves + twelve gather waves a 128-channel layer gets 12 channels per wave and 36 accumulator pairs =
// a 1x3 tile (5x7 window: 11.7 table builds per output position for a 5x5 kernel), an eleventh of the gather lanes idle.
// Here a wave gathers 8 channels (16 x 8 = 128, no idle lane) for a 2x2 tile (32 pairs; 6x6 window: 9.0 builds per
// position) and builds 4 of the 64 result tiles of every stage: image tile it = wave >> 1, row tiles 4 (wave & 1) .. + 3.
// Same table entries, same (kh, kw, m) order per output as k_conv_aprx: bit-identical results.
// Per stage period: [multiply stage s + 1 into the other buffer] [wave 0: DMA the program row of stage s + 2]
// [operand loads of stage s + 2] [pick the block of stage s + 1] [gather stage s] [barrier].  The DMA is issued BEFORE the
// operand loads, so that the barrier's counted vmcnt waits for it and leaves the operand loads in flight.
// K = 128, MFMA builder, Cs = 4 KS dims in every sub-space, program table of the (8 channels per wave, 2x2) layout —
// two channel chunks of the 12-wave table layout are one workgroup here (waves 0-11: chunk 0, 12-15: chunk 1).
// ------------------------------------------------------------------------------------------------
template <int KS>
struct SymOps {
  float a[4][KS];          // code-book operand of the wave's four row tiles
  float b[KS];             // activation operand of the wave's image tile
};
template <int KS>
__device__ __forceinline__ void sym_load(SymOps<KS>& o, const char* __restrict__ xbase, uint32_t xoff0, uint32_t bLane,
                                         const float* __restrict__ ctrd, int Cs, int m, uint32_t laneA, int rt0) {
  constexpr int K = 128;
  const float* __restrict__ cbU = ctrd + (size_t)m * Cs * K + rt0 * 16;            // uniform
  const char* __restrict__ xbU = xbase + xoff0 + (uint32_t)(m * Cs) * XROWB;       // uniform
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o.a[i][ks] = (cbU + (ks * 4 * K + i * 16))[laneA];
    o.b[ks] = *reinterpret_cast<const float*>(xbU + (uint32_t)(ks * 4) * XROWB + bLane);
  }
}
template <int KS>
__device__ __forceinline__ f32x4 sym_tile(const SymOps<KS>& o, int i) {
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 c = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][0], o.b[0], zero, 0, 0, 0);
  if (KS > 1) c = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][KS - 1], o.b[KS - 1], c, 0, 0, 0);
  return c;
}
// m0v = LDS byte address of (buffer, image tile, first row tile of the wave); the four tiles go out in pairs, the stores of
// a pair behind the matrix instructions of the next
template <int KS>
__device__ __forceinline__ void sym_store(const SymOps<KS>& o, uint32_t m0v) {
  const f32x4 v0 = sym_tile<KS>(o, 0);
  const f32x4 v1 = sym_tile<KS>(o, 1);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v2 = sym_tile<KS>(o, 2);
  store_tile_lo<0>(v0, m0v); store_tile_hi<0>(v0, m0v);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v3 = sym_tile<KS>(o, 3);
  store_tile_lo<1>(v1, m0v); store_tile_hi<1>(v1, m0v);
  __builtin_amdgcn_sched_barrier(0);
  store_tile_lo<2>(v2, m0v); store_tile_hi<2>(v2, m0v);
  store_tile_lo<3>(v3, m0v); store_tile_hi<3>(v3, m0v);
}

template <int KS>
__global__ __launch_bounds__(NW * 64) void k_conv_sym(ConvParams p, int tilesX, int tilesY) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int TH = 2, TW = 2, CPW = 8, NP = 4, HC = 4;
  constexpr int DW = idx_dwords(CPW);
  constexpr int NB = (NP * DW + 3) / 4 * 4;            // dwords of one wave half's program block
  constexpr int ROWB = NW * 2 * NB * 4;                // program bytes of the workgroup's 16 waves per entry (contiguous)
  constexpr int WGROW12 = NGW * 2 * NB * 4;            // bytes of ONE channel chunk's row in the 12-wave table layout
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int rank = (int)(blockIdx.x / (unsigned)p.panels), panel = (int)(blockIdx.x % (unsigned)p.panels);
  int ty, tx;
  tile_of_rank(rank, tilesY, tilesX, ty, tx);
  const int grp = blockIdx.y;
  const int Cg = p.Cin / p.grp, Ctg = p.Ct / p.grp;   // Ctg == 128
  const int M = p.M;
  const int ho0 = ty * TH, wo0 = tx * TW;
  const int hoL = min(ho0 + TH, p.Ho) - 1, woL = min(wo0 + TW, p.Wo) - 1;
  const int hiL = max(0, ho0 * p.stride - p.pad), hiU = min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1);
  ConvGeom g;
  g.W = p.W; g.Cin = p.Cin; g.knl = p.knl; g.M = M; g.G = 1; g.rowStride = 0;
  g.pixStride = (uint32_t)p.Cin * (uint32_t)XROWB;
  g.MG = M;
  g.wiL = max(0, wo0 * p.stride - p.pad);
  g.wiU = min(p.W - 1, woL * p.stride - p.pad + p.knl - 1);
  g.slide = 0; g.hiL = hiL; g.hiU = hiU; g.period = 1;
  const int cols = g.wiU - g.wiL + 1;
  const int S = (hiU - hiL + 1) * cols * g.MG;
  const int Sp = (S + 1) & ~1;
  const StagePos first = {hiL, g.wiL, 0, 0};
  if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();

  // ---- builder side of this wave
  const int it = wave >> 1, rt0 = (wave & 1) * 4, bwS = it >> 1;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t laneA = lk * 128 + (li ^ ((uint32_t)bwS << 2));      // rows pre-swizzled for the tile's slot order (mfma_load)
  const uint32_t bLane = lk * XROWB + (uint32_t)it * 64 + li * 4;
  const char* __restrict__ xbase =
      reinterpret_cast<const char*>(p.src + ((size_t)panel * p.H * p.W * p.Cin + (size_t)grp * Cg) * PANEL);
  const uint32_t m0buf0 = (uint32_t)it * TILEB + (uint32_t)rt0 * 1024u, m0buf1 = m0buf0 + STAGE_BYTES;
  const int Cs = p.Cs;

  // ---- gather side
  const int half = lane >> 5, quad = lane & 31;
  const int cw0 = wave * CPW;
  const int cl0 = cw0 + half * HC;
  const uint32_t laneLds = (uint32_t)(quad >> 2) * TILEB | (uint32_t)(quad >> 3) * 64 | (uint32_t)(quad & 3) * 16;
  f32x2 acc[NP][CPW];
  {
    const float* __restrict__ bp = p.bias + grp * Ctg + cl0;
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const float b = bp[j];
#pragma unroll
      for (int q = 0; q < NP; ++q) { acc[q][2 * j] = f32x2{b, b}; acc[q][2 * j + 1] = f32x2{b, b}; }
    }
  }
  int rowStart[TH], colStart[TW];
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) rowStart[dy] = (ho0 + dy < p.Ho) ? (ho0 + dy) * p.stride - p.pad : -(1 << 28);
#pragma unroll
  for (int dx = 0; dx < TW; ++dx) colStart[dx] = (wo0 + dx < p.Wo) ? (wo0 + dx) * p.stride - p.pad : -(1 << 28);
  const int rfW = (TW - 1) * p.stride + p.knl;
  const int ry0 = ho0 * p.stride - p.pad, rx0 = wo0 * p.stride - p.pad;
  const uint32_t entryB = (uint32_t)(p.grp * 2) * WGROW12;              // two channel chunks per group in the table
  const char* __restrict__ progWg = reinterpret_cast<const char*>(p.progS) + (size_t)(grp * 2) * WGROW12;
  auto rowOf = [&](const StagePos& q, int idx) {
    const StagePos c = (idx < S) ? q : first;
    return progWg + (size_t)(uint32_t)(((c.hi - ry0) * rfW + (c.wi - rx0)) * M + c.mg) * entryB;
  };
  auto posOf = [&](const StagePos& q, int idx) { return (idx < S) ? q : first; };
  const uint32_t myBlk = (uint32_t)(wave * 2 + half) * NB * 4;
  const bool loader = wave == 0;

  SymOps<KS> ops;
  IdxBlk<NB> ba, bb;
  StagePos c0 = first;
  StagePos c1 = next_pos(c0, g);
  StagePos c2 = next_pos(c1, g);
  StagePos c3 = next_pos(c2, g);
  // Program rows: THREE LDS buffers, the row of stage t in buffer t % 3, fetched by DMA two periods before it is read — wave
  // 0 never waits for its DMA at a barrier: it has landed when the operands loaded after it are consumed a period later.
  uint32_t rb0 = IDX_LDS, rb1 = IDX_LDS + IDX_BUF, rb2 = IDX_LDS + 2 * IDX_BUF;   // buffers of stages s, s + 1, s + 2 (= s + 3)
  // prologue: stage 0 -> buffer 0; operands of stage 1; block of stage 0 from HBM, rows of stages 1 and 2 by DMA
  sym_load<KS>(ops, xbase, pixel_off(c0, g), bLane, p.ctrd, Cs, c0.mg, laneA, rt0);
  sym_store<KS>(ops, m0buf0);
  {
    const StagePos q = posOf(c1, 1);
    sym_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd, Cs, q.mg, laneA, rt0);
  }
  blk_load(ba, rowOf(c0, 0) + myBlk);
  if (loader) { idx_row_to_lds<ROWB>(rowOf(c1, 1), rb1, lane); idx_row_to_lds<ROWB>(rowOf(c2, 2), rb2, lane); }
  barrier_after_lds_dma();
  for (int s = 0; s < Sp; s += 2) {
    // ---- period s: stage s + 1 -> buffer 1, gather stage s out of buffer 0
    sym_store<KS>(ops, m0buf1);
    __builtin_amdgcn_sched_barrier(0);
    if (loader) idx_row_to_lds<ROWB>(rowOf(c3, s + 3), rb0, lane);      // row of stage s + 3 into the buffer stage s has left
    __builtin_amdgcn_sched_barrier(0);
    {
      const StagePos q = posOf(c2, s + 2);
      sym_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd, Cs, q.mg, laneA, rt0);
    }
    blk_load(bb, lds + rb1 + myBlk);                                    // stage s + 1
    conv_gather_prog<TH, TW, CPW, NB>(acc, ba, c0, g, rowStart, colStart, laneLds, 1);
    c0 = c1; c1 = c2; c2 = c3; c3 = next_pos(c3, g);
    { const uint32_t t = rb0; rb0 = rb1; rb1 = rb2; rb2 = t; }
    barrier_after_lds_writes();
    // ---- period s + 1: stage s + 2 -> buffer 0, gather stage s + 1 out of buffer 1
    sym_store<KS>(ops, m0buf0);
    __builtin_amdgcn_sched_barrier(0);
    if (loader) idx_row_to_lds<ROWB>(rowOf(c3, s + 4), rb0, lane);
    __builtin_amdgcn_sched_barrier(0);
    {
      const StagePos q = posOf(c2, s + 3);
      sym_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd, Cs, q.mg, laneA, rt0);
    }
    blk_load(ba, lds + rb1 + myBlk);                                    // stage s + 2
    conv_gather_prog<TH, TW, CPW, NB>(acc, bb, c0, g, rowStart, colStart, laneLds | STAGE_BYTES, in_range(s + 1, S));
    c0 = c1; c1 = c2; c2 = c3; c3 = next_pos(c3, g);
    { const uint32_t t = rb0; rb0 = rb1; rb1 = rb2; rb2 = t; }
    barrier_after_lds_writes();
  }
  // ---- results
  float* __restrict__ dst = p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int ho = ho0 + q / TW, wo = wo0 + q % TW;
    if (ho < p.Ho && wo < p.Wo) {
      float* o = dst + ((size_t)(ho * p.Wo + wo) * p.Ct + grp * Ctg + cl0) * PANEL + 4 * quad;
#pragma unroll
      for (int j = 0; j < HC; ++j) {
        f32x4 v = {acc[q][2 * j].x, acc[q][2 * j].y, acc[q][2 * j + 1].x, acc[q][2 * j + 1].y};
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (0.0f < v[e]) ? v[e] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(o + j * PANEL) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fully connected: stages of G sub-spaces; 4 builder waves + 12 gather waves x CPW channels, as in the
// conv kernel; the gather waves walk the sub-spaces with the offsets of the next two always in flight.
// Optional split over the sub-space axis (blockIdx.z): partial sums go to p.partial and are reduced by
// k_sum_partials.
// ------------------------------------------------------------------------------------------------
template <int CPW, int KT, int KS>
__global__ __launch_bounds__(NW * 64) void k_fc_aprx(FcParams p, int G, int stagesPerSplit, int rowStride) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int KTT = KT > 0 ? KT : 1;
  constexpr int HC = CPW / 2;
  constexpr int DW = idx_dwords(CPW);
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int panel = blockIdx.y;
  const int split = blockIdx.z;
  const int M = p.M;
  const int mBeg = split * stagesPerSplit * G;
  const int mEnd = min(M, mBeg + stagesPerSplit * G);
  const int S = (mEnd - mBeg + G - 1) / G;
  const int Sp = (S + 1) & ~1;                      // every wave runs Sp stage periods (barriers)
  if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();   // see k_conv_aprx

  const WaveRole role = assign_roles(lds, wave, lane);
  if (role.builder) {
    const int bw = uni(role.idx);                   // kept in an SGPR: it feeds the M0 operand of the add-TID stores
    const int K = p.K, Cs = p.Cs;
    __builtin_amdgcn_s_setprio(QCNN_PRIO_BUILDER);
    const char* __restrict__ xbase = reinterpret_cast<const char*>(p.src + (size_t)panel * p.D * PANEL);
    const XAddr xa = xaddr_panel(bw, lane);
    MfmaOps<KTT, KS> opsA, opsB;                          // two operand sets, see k_conv_aprx
    const int mLastStage = mBeg + max(S - 1, 0) * G;  // operand prefetches past the end re-fetch the last stage
    if (S > 0) {
      if (KT > 0) {
        mfma_load<KTT, KS>(opsA, xbase, 0u, xa, p.ctrd, Cs, mBeg, bw, lane);
        mfma_store<KTT, KS, 0>(opsA, Cs, p.D, mBeg, mEnd, bw, lane, p.lutF16);
        mfma_load<KTT, KS>(opsA, xbase, 0u, xa, p.ctrd, Cs, min(mBeg + G, mLastStage), bw, lane);
        __builtin_amdgcn_sched_barrier(0);
        mfma_load<KTT, KS>(opsB, xbase, 0u, xa, p.ctrd, Cs, min(mBeg + 2 * G, mLastStage), bw, lane);
      } else {
        build_stage_exact(lds, xbase, 0u, xa, p.ctrd, K, Cs, p.D, G, mBeg, mEnd, bw, lane, p.pd);
      }
    }
    barrier_after_lds_writes();
    for (int s = 0; s < Sp; s += 2) {                  // straight-line body, see k_conv_aprx
      const int m0 = mBeg + s * G;
      if (KT > 0) {
        mfma_store<KTT, KS, 1>(opsA, Cs, p.D, min(m0 + G, mLastStage), mEnd, bw, lane, p.lutF16);
        mfma_load<KTT, KS>(opsA, xbase, 0u, xa, p.ctrd, Cs, min(m0 + 3 * G, mLastStage), bw, lane);
      } else if (s + 1 < S) {
        build_stage_exact(lds + STAGE_BYTES, xbase, 0u, xa, p.ctrd, K, Cs, p.D, G, m0 + G, mEnd, bw, lane, p.pd);
      }
      barrier_after_lds_writes();
      if (KT > 0) {
        mfma_store<KTT, KS, 0>(opsB, Cs, p.D, min(m0 + 2 * G, mLastStage), mEnd, bw, lane, p.lutF16);
        mfma_load<KTT, KS>(opsB, xbase, 0u, xa, p.ctrd, Cs, min(m0 + 4 * G, mLastStage), bw, lane);
      } else if (s + 2 < S) {
        build_stage_exact(lds, xbase, 0u, xa, p.ctrd, K, Cs, p.D, G, m0 + 2 * G, mEnd, bw, lane, p.pd);
      }
      barrier_after_lds_writes();
    }
    return;
  }

  const int gw = role.idx;
  const int half = lane >> 5, quad = lane & 31;
  const int cw0 = (blockIdx.x * NGW + gw) * CPW;
  const bool active = cw0 < p.Ct;
  const int cl0 = cw0 + half * HC;
  constexpr int BW = idx_bytes_dwords(DW);
  const uint8_t* __restrict__ rowsW = p.rows + (size_t)(blockIdx.x * NGW + gw) * 2 * (4 * BW);
  const uint32_t laneOff = (uint32_t)half * (4 * BW);
  // XOR-ed with a pre-scaled row offset it gives the lane's read address: tile, slot swizzle (tile >> 1), 16-byte quarter
  const uint32_t laneLds = (uint32_t)(quad >> 2) * TILEB | (uint32_t)(quad >> 3) * 64 | (uint32_t)(quad & 3) * 16;

  f32x2 acc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) acc[c] = f32x2{0.0f, 0.0f};
  if (split == 0) {
    const float* __restrict__ bp = p.bias + (active ? cl0 : 0);   // over-read stays inside the arena
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const float b = bp[j];
      acc[2 * j] = f32x2{b, b}; acc[2 * j + 1] = f32x2{b, b};
    }
  }

  // sub-space stream: two offset sets alternate, each refilled (two sub-spaces ahead) right after its use; a
  // barrier closes a stage every G sub-spaces (a pair may straddle it).  Sub-spaces past mEnd (ragged last
  // stage) are skipped inside the look-up blocks; the stream is padded to Sp stages so that every wave meets
  // the same barriers.
  IdxB<BW> ia, ib;
  const int activeI = in_range(cw0, p.Ct);             // `active` as an integer (see in_range)
  const int mClamp = max(mEnd - 1, mBeg);
  vload_idx(ia, rowsW + (size_t)mBeg * rowStride, laneOff);
  vload_idx(ib, rowsW + (size_t)min(mBeg + 1, mClamp) * rowStride, laneOff);
  __builtin_amdgcn_s_setprio(QCNN_PRIO_GATHER);
  barrier_plain();
  {
    int r = 0;                      // sub-spaces of the current stage already consumed
    uint32_t stage = laneLds;
    const int T = Sp * G;           // sub-space slots incl. padding (even)
    for (int k = 0; k < T; k += 2) {
      const int m = mBeg + k;
      gather_apply<CPW>(acc, expand_idx<DW>(ia), stage, activeI & in_range(m, mEnd));
      vload_idx(ia, rowsW + (size_t)min(m + 2, mClamp) * rowStride, laneOff);
      if (++r == G) { r = 0; stage ^= STAGE_BYTES; barrier_plain(); }
      gather_apply<CPW>(acc, expand_idx<DW>(ib), stage, activeI & in_range(m + 1, mEnd));
      vload_idx(ib, rowsW + (size_t)min(m + 3, mClamp) * rowStride, laneOff);
      if (++r == G) { r = 0; stage ^= STAGE_BYTES; barrier_plain(); }
    }
  }

  if (active) {
    float* base = (p.msplit > 1) ? p.partial + (size_t)split * p.panels * p.Ct * PANEL : p.dst;
    float* o = base + ((size_t)panel * p.Ct + cl0) * PANEL + 4 * quad;
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      if (cl0 + j < p.Ct) {
        f32x4 v = {acc[2 * j].x, acc[2 * j].y, acc[2 * j + 1].x, acc[2 * j + 1].y};
        if (p.relu && p.msplit == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (0.0f < v[e]) ? v[e] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(o + j * PANEL) = v;
      }
    }
  }
}

// .cbn payload -> row-offset table: one thread per assignment.  Element e of the file order
// [Ct][taps][M] sits in block e / per at bit (e % per) * bits, MSB first (include/FileIO.h:128-166; values never
// straddle a block); it lands at [tap][m][slot entry of its channel] as the pre-scaled LDS offset of its stage row
// (the same value qcnn_model_set_layer_params computes on the host).
__global__ void k_decode_cbn(const uint8_t* __restrict__ blocks, int bits, size_t n, int Ct, int taps, int M, int K,
                             int G, QkSlots sl, uint8_t* __restrict__ rows, int* __restrict__ bad) {
  const int per = 4096 * 8 / bits;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = e / per;
    const int bit0 = (int)(e % per) * bits;
    const uint8_t* b = blocks + blk * 4096 + (bit0 >> 3);
    const unsigned w = ((unsigned)b[0] << 8) | (unsigned)b[(bit0 & 7) + bits > 8 ? 1 : 0];   // <= 8 bits: at most two bytes
    const unsigned v = (w >> (16 - (bit0 & 7) - bits)) & ((1u << bits) - 1u);
    const int m = (int)(e % M);
    const size_t ct = e / M;
    const int t = (int)(ct % taps), ch = (int)(ct / taps);
    if ((int)v >= K) { atomicOr(bad, 1); continue; }
    const int entry = qk_slot_entry(sl, ch / sl.C, ch % sl.C);
    rows[((size_t)t * M + m) * sl.rowStride + entry] = (uint8_t)qcnn_row_slot((m % G) * K + (int)v);
  }
}

#ifdef QCNN_TRACE
}  // namespace
extern "C" int qcnn_debug_trace_read(unsigned long long* host, int block) {
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(qcnn_trace_buf), sizeof(unsigned long long) * (16 * 64 * 2 + 16 + 16 * 64));
  if (e != hipSuccess) return 1;
  e = hipMemcpyToSymbol(HIP_SYMBOL(qcnn_trace_block), &block, sizeof(int));
  return e == hipSuccess ? 0 : 1;
}
namespace {
#endif


// rows (plain table of row slots, [kh][kw][M][rowStride]) -> program table of pre-scaled offsets ([ry][rx][M][rowU16], QkProgram): one thread per entry
__global__ __launch_bounds__(256) void k_build_program(const uint8_t* __restrict__ rows, uint16_t* __restrict__ prog,
                                                       QkSlots src, QkSlots sl, QkProgram pg, int knl, int stride, int M, size_t n,
                                                       int slide) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e % (size_t)pg.rowU16);
    const int row = (int)(e / (size_t)pg.rowU16);
    const int m = row % M, pix = row / M;
    const int ry = pix / pg.rfW, rx = pix % pg.rfW;
    const int wh = r / pg.blkU16, r3 = r % pg.blkU16;       // (workgroup slice, wave, half) and the place inside its block
    const int pos = r3 / sl.hp, j = r3 % sl.hp;
    uint16_t v = 0;
    if (pos < pg.np && j < sl.cpw / 2) {
      // tile kernel: position (dy, dx) looks at tap (ry - dy * stride, rx - dx * stride); sliding: slot `pos` at tap row
      // (ry - pos * stride) modulo the period rfH = slots * stride, tap column rx
      // (slide: pg.tw = slots per column, positions are [column][slot])
      const int kh = slide ? ((ry - (pos % pg.tw) * stride) % pg.rfH + pg.rfH) % pg.rfH : ry - (pos / pg.tw) * stride;
      const int kw = slide ? rx - (pos / pg.tw) * stride : rx - (pos % pg.tw) * stride;
      // the channel this entry belongs to in the destination's wave split, and where the source table keeps it
      const int half = wh & 1, wave = (wh >> 1) % (sl.chunks * QCNN_GATHER_WAVES), g = (wh >> 1) / (sl.chunks * QCNN_GATHER_WAVES);
      const int ch = wave * sl.cpw + half * (sl.cpw / 2) + j;
      const int at = qk_slot_entry(src, g, ch);
      if ((unsigned)kh < (unsigned)knl && (unsigned)kw < (unsigned)knl && at >= 0)
        v = (uint16_t)(rows[(size_t)((kh * knl + kw) * M + m) * src.rowStride + at] * 64);
    }
    prog[e] = v;
  }
}

// Split tiles (ConvParams::splitZ): dst = sum over the slices, in slice order, of the partial sums; optional ReLU.
// grid (tail tiles * positions per tile, panels, chunks of 1024 float4), 256 threads x 4 float4.
__global__ __launch_bounds__(256) void k_conv_sum(const f32x4* __restrict__ partial, f32x4* __restrict__ dst, int splitFrom,
                                                  int Z, int panels, int tilesX, int tilesY, int TH, int TW, int Ho, int Wo,
                                                  int Ct, int relu) {
  const int NP = TH * TW;
  const int tailTile = blockIdx.x / NP, q = blockIdx.x % NP, panel = blockIdx.y;
  int ty, tx;
  tile_of_rank(splitFrom + tailTile, tilesY, tilesX, ty, tx);
  const int ho = ty * TH + q / TW, wo = tx * TW + q % TW;
  if (ho >= Ho || wo >= Wo) return;
  const size_t rowQuads = (size_t)Ct * (PANEL / 4);                       // float4 of one position
  const f32x4* __restrict__ src = partial + (((size_t)tailTile * Z * panels + panel) * NP + q) * rowQuads;
  const size_t sliceStride = (size_t)panels * NP * rowQuads;
  f32x4* __restrict__ out = dst + ((size_t)panel * Ho * Wo + (size_t)ho * Wo + wo) * rowQuads;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t e = ((size_t)blockIdx.z * 4 + k) * 256 + threadIdx.x;
    if (e < rowQuads) {
      f32x4 v = src[e];
      for (int z = 1; z < Z; ++z) {
        const f32x4 w = src[z * sliceStride + e];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = __fadd_rn(v[c], w[c]);
      }
      if (relu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (0.0f < v[c]) ? v[c] : 0.0f;
      }
      out[e] = v;
    }
  }
}

template <int TH, int TW, int CPW>
hipError_t launch_conv(const ConvParams& p, const QkSlots& sl, int lutMode, hipStream_t st) {
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH;
  const int tiles = tilesX * tilesY;
  const bool split = p.splitZ > 1 && p.splitFrom >= 0 && p.splitFrom < tiles && p.partial != nullptr;
  ConvParams q = p;
  if (!split) { q.splitFrom = tiles; q.splitZ = 1; q.partial = nullptr; }
  const int tailTiles = tiles - q.splitFrom;
  const dim3 grid((q.splitFrom + tailTiles * q.splitZ) * p.panels, sl.chunks * p.grp, 1);
  const size_t shm = (size_t)2 * STAGE_BYTES + 2 * IDX_BUF;
  const int G = qcnn_stage_group(p.K);
  const bool two = min(p.Cin / p.grp, p.Cs) > 4;      // MFMA k-steps (4 dims each) that carry data
  auto kern = k_conv_aprx<TH, TW, CPW, 0, 1>;
  if (lutMode == 1 && p.K == 128) kern = two ? k_conv_aprx<TH, TW, CPW, 8, 2> : k_conv_aprx<TH, TW, CPW, 8, 1>;
  if (lutMode == 1 && p.K == 64) kern = two ? k_conv_aprx<TH, TW, CPW, 4, 2> : k_conv_aprx<TH, TW, CPW, 4, 1>;
  if (lutMode == 1 && p.K == 32) kern = two ? k_conv_aprx<TH, TW, CPW, 2, 2> : k_conv_aprx<TH, TW, CPW, 2, 1>;
  if (lutMode == 1 && p.K == 16) kern = two ? k_conv_aprx<TH, TW, CPW, 1, 2> : k_conv_aprx<TH, TW, CPW, 1, 1>;
  hipError_t e = allow_big_lds(reinterpret_cast<const void*>(kern), (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, q, tilesX, tilesY, sl.chunks, G, sl.rowStride);
  e = hipGetLastError();
  if (e != hipSuccess || !split) return e;
  return qk_conv_sum(q.partial, p.dst, q.splitFrom, q.splitZ, p.panels, tilesX, tilesY, TH, TW, p.Ho, p.Wo, p.Ct, p.relu, st);
}

// sliding variant: grid.x = (segments x output columns, longest segments first) x panels
template <int NC, int NS, int CPW>
hipError_t launch_conv_slide(const ConvParams& p, const QkSlots& sl, int lutMode, hipStream_t st) {
  const dim3 grid((unsigned)(p.nSeg * ((p.Wo + NC - 1) / NC) * p.panels), sl.chunks * p.grp, 1);
  const size_t shm = (size_t)2 * STAGE_BYTES + 2 * IDX_BUF;
  const bool two = min(p.Cin / p.grp, p.Cs) > 4;
  auto kern = two ? k_conv_aprx<NC, NS, CPW, 8, 2, true> : k_conv_aprx<NC, NS, CPW, 8, 1, true>;
  hipError_t e = allow_big_lds(reinterpret_cast<const void*>(kern), (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, 0, 0, sl.chunks, 1, sl.rowStride);
  return hipGetLastError();
}

template <int CPW>
hipError_t launch_fc(const FcParams& p, const QkSlots& sl, int lutMode, hipStream_t st) {
  const int G = qcnn_stage_group(p.K);
  const int stages = (p.M + G - 1) / G;
  const int stagesPerSplit = (stages + p.msplit - 1) / p.msplit;
  const dim3 grid(sl.chunks, p.panels, p.msplit);
  const size_t shm = (size_t)2 * STAGE_BYTES;
  const bool two = min(p.D, p.Cs) > 4;
  auto kern = k_fc_aprx<CPW, 0, 1>;
  if (lutMode == 1 && p.K == 128) kern = two ? k_fc_aprx<CPW, 8, 2> : k_fc_aprx<CPW, 8, 1>;
  if (lutMode == 1 && p.K == 64) kern = two ? k_fc_aprx<CPW, 4, 2> : k_fc_aprx<CPW, 4, 1>;
  if (lutMode == 1 && p.K == 32) kern = two ? k_fc_aprx<CPW, 2, 2> : k_fc_aprx<CPW, 2, 1>;
  if (lutMode == 1 && p.K == 16) kern = two ? k_fc_aprx<CPW, 1, 2> : k_fc_aprx<CPW, 1, 1>;
  hipError_t e = allow_big_lds(reinterpret_cast<const void*>(kern), (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, G, stagesPerSplit, sl.rowStride);
  return hipGetLastError();
}

// symmetric workgroups (k_conv_sym): grid.x = 2x2 tiles (heaviest first) x panels, grid.y = groups
hipError_t launch_conv_sym(const ConvParams& p, hipStream_t st) {
  const int tilesX = (p.Wo + 1) / 2, tilesY = (p.Ho + 1) / 2;
  const dim3 grid((unsigned)(tilesX * tilesY * p.panels), (unsigned)p.grp, 1);
  const size_t shm = (size_t)2 * STAGE_BYTES + 3 * IDX_BUF;
  const bool two = min(p.Cin / p.grp, p.Cs) > 4;
  auto kern = two ? k_conv_sym<2> : k_conv_sym<1>;
  hipError_t e = allow_big_lds(reinterpret_cast<const void*>(kern), (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, tilesX, tilesY);
  return hipGetLastError();
}

}  // namespace

hipError_t qk_conv_sym(const ConvParams& p, hipStream_t st) {
  if (!qk_conv_sym_shape(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K) || p.progS == nullptr || p.srcNchw) return hipErrorInvalidValue;
  return launch_conv_sym(p, st);
}

hipError_t qk_conv_aprx(const ConvParams& pIn, int lutMode, hipStream_t st) {
  ConvParams p = pIn;
  p.lutF16 = (lutMode >= 2) ? 1 : 0;        // 2, 3: entries rounded to fp16, kept in f32 slots, fp32 sums (layers without an fp16 form)
  if (lutMode >= 2) lutMode = 1;
  if (p.pd < 1 || (p.pd > 1 && lutMode != 0)) return hipErrorInvalidValue;   // pseudo sub-spaces: the exact-builder kernels only
  const int Ctg = p.Ct / p.grp;
  if (Ctg % 2 || p.Cs > QCNN_MAX_CS || p.K > QCNN_MAX_K) return hipErrorInvalidValue;
  const QkSlots sl = qk_conv_slots(Ctg, p.grp);
  if (p.nSeg > 0 && p.progS != nullptr && p.K == 128 && lutMode != 0) {      // sliding variant (qk_conv_plan_slide)
    const QkSlide sc = qk_slide_config(Ctg, p.grp, p.knl, p.stride);
    switch (sc.nc * 1000 + sc.ns * 100 + sc.sl.cpw) {
      case 1216: return launch_conv_slide<1, 2, 16>(p, sc.sl, lutMode, st);
      case 1212: return launch_conv_slide<1, 2, 12>(p, sc.sl, lutMode, st);
      case 1208: return launch_conv_slide<1, 2, 8>(p, sc.sl, lutMode, st);
      case 1312: return launch_conv_slide<1, 3, 12>(p, sc.sl, lutMode, st);
      case 1308: return launch_conv_slide<1, 3, 8>(p, sc.sl, lutMode, st);
      case 1306: return launch_conv_slide<1, 3, 6>(p, sc.sl, lutMode, st);
      case 1304: return launch_conv_slide<1, 3, 4>(p, sc.sl, lutMode, st);
      case 1408: return launch_conv_slide<1, 4, 8>(p, sc.sl, lutMode, st);
      case 1506: return launch_conv_slide<1, 5, 6>(p, sc.sl, lutMode, st);
      case 1504: return launch_conv_slide<1, 5, 4>(p, sc.sl, lutMode, st);
      case 2306: return launch_conv_slide<2, 3, 6>(p, sc.sl, lutMode, st);      // two-column strips: six slots of <= 6 channels
      case 2304: return launch_conv_slide<2, 3, 4>(p, sc.sl, lutMode, st);
      default: break;
    }
  }
  switch (sl.cpw) {
    case 32: return launch_conv<1, 1, 32>(p, sl, lutMode, st);   // 1 position  x 12 x 32 channels
    case 24: return launch_conv<1, 1, 24>(p, sl, lutMode, st);   // 1 position  x 12 x 24
    case 16: return launch_conv<1, 2, 16>(p, sl, lutMode, st);   // 2 positions x 12 x 16
    case 12: return launch_conv<1, 3, 12>(p, sl, lutMode, st);   // 3 positions x 12 x 12
    case 8: return launch_conv<2, 2, 8>(p, sl, lutMode, st);     // 4 positions x 12 x 8
    case 6: return launch_conv<2, 3, 6>(p, sl, lutMode, st);     // 6 positions x 12 x 6
    default: return launch_conv<2, 4, 4>(p, sl, lutMode, st);    // 8 positions x 12 x 4
  }
}

int qk_fc_channels_per_block(int Ct) { return NGW * qk_fc_slots(Ct).cpw; }

// p.msplit is chosen by the caller (engine): 1 keeps the reference's summation order.
hipError_t qk_fc_aprx(const FcParams& pIn, int lutMode, hipStream_t st) {
  FcParams p = pIn;
  p.lutF16 = (lutMode >= 2) ? 1 : 0;
  if (lutMode >= 2) lutMode = 1;
  if (p.pd < 1 || (p.pd > 1 && lutMode != 0)) return hipErrorInvalidValue;
  if (p.Ct % 2 || p.Cs > QCNN_MAX_CS || p.K > QCNN_MAX_K || p.msplit < 1) return hipErrorInvalidValue;
  const QkSlots sl = qk_fc_slots(p.Ct);
  switch (sl.cpw) {
    case 32: return launch_fc<32>(p, sl, lutMode, st);
    case 8: return launch_fc<8>(p, sl, lutMode, st);
    default: return launch_fc<4>(p, sl, lutMode, st);
  }
}

hipError_t qk_decode_cbn(const uint8_t* blocks, int bits, size_t n, int Ct, int taps, int M, int K, QkSlots sl,
                         uint8_t* rows, int* bad, hipStream_t st) {
  if (bits < 1 || bits > 8) return hipErrorInvalidValue;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(k_decode_cbn, dim3(grid ? grid : 1), dim3(256), 0, st, blocks, bits, n, Ct, taps, M, K,
                     qcnn_stage_group(K), sl, rows, bad);
  return hipGetLastError();
}

hipError_t qk_build_program(const uint8_t* rows, uint16_t* prog, QkSlots src, QkSlots dst, QkProgram pg, int knl, int stride,
                            int M, hipStream_t st, int slide) {
  const size_t n = (size_t)pg.rfH * pg.rfW * M * pg.rowU16;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(k_build_program, dim3(grid ? grid : 1), dim3(256), 0, st, rows, prog, src, dst, pg, knl, stride, M, n, slide);
  return hipGetLastError();
}

hipError_t qk_conv_sum(const float* partial, float* dst, int splitFrom, int Z, int panels, int tilesX, int tilesY, int TH, int TW, int Ho,
                       int Wo, int Ct, int relu, hipStream_t st) {
  const int tailTiles = tilesX * tilesY - splitFrom;
  const int rowQuads = Ct * (PANEL / 4);
  hipLaunchKernelGGL(k_conv_sum, dim3(tailTiles * TH * TW, panels, (rowQuads + 1023) / 1024), dim3(256), 0, st,
                     reinterpret_cast<const f32x4*>(partial), reinterpret_cast<f32x4*>(dst), splitFrom, Z, panels, tilesX, tilesY, TH, TW,
                     Ho, Wo, Ct, relu);
  return hipGetLastError();
}
