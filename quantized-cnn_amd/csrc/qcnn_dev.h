// qcnn_dev.h — device-side building blocks shared by the table kernels (qcnn_kernels.hip: k_conv_aprx / k_conv_sym / k_fc_aprx;
// qcnn_sym8.hip: k_conv_sym8): LDS stage geometry, the look-up block macros, the add-TID stores of a result tile, the stage
// sequence of a conv tile, program rows by LDS-DMA.  Internal; everything lives in an anonymous namespace per translation unit.
#ifndef QCNN_DEV_H_
#define QCNN_DEV_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "qcnn_kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PANEL = QCNN_PANEL;              // images per panel

constexpr int TILEB = QCNN_TILE_BYTES;         // LDS bytes of one image tile of a stage

constexpr int STAGE_BYTES = QCNN_STAGE_BYTES;  // 64 KB; two stages = 128 KB of the 160 KB LDS

constexpr int XROWB = PANEL * 4;               // bytes of one activation row in HBM

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Workgroup barrier WITHOUT the implicit "wait for everything" of __syncthreads(): the builder waves
// wait for their LDS writes only (their operand prefetch of the stage after next stays in flight), the
// gather waves wait for nothing (their look-ups were consumed by the adds; their index prefetch stays
// in flight).  The "memory" clobber keeps the compiler from moving LDS accesses across it.
__device__ __forceinline__ void barrier_after_lds_writes() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void barrier_plain() { asm volatile("s_barrier" ::: "memory"); }

#define Q_AD(a, w, sel) "v_xor_b32_sdwa " a ", %[" w "], %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" sel " src1_sel:DWORD\n\t"
#define Q_RD(v, a) "ds_read_b128 " v ", " a "\n\t"
#define Q_ACC(n, c0, c1, lo, hi) \
  "s_waitcnt lgkmcnt(" n ")\n\tv_pk_add_f32 %[" c0 "], " lo ", %[" c0 "]\n\tv_pk_add_f32 %[" c1 "], " hi ", %[" c1 "]\n\t"
#define Q_SKIP "s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lqskip%=\n\t"
#define Q_CLOB8 "scc", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",     \
                "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120",    \
                "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define Q_CLOB4 "scc", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",   \
                "v124", "v125", "v126", "v127"

// LDS slot of a stage row (device copy of qcnn_row_slot)
__device__ __forceinline__ int row_slot(int r) { return (r & 0x70) | ((r & 3) << 2) | ((r >> 2) & 3); }

// Result tile (image tile `it` of this wave, row tile I) -> LDS: element e of the four result registers goes
// to slots 16I + 4e .. 4e+3 of the tile, 256 contiguous bytes, by ONE ds_write_addtid_b32 (address = M0[15:0] +
// 16-bit offset + 4 * lane: no address register, half the LDS-path cycles of ds_write_b32).  The position of a
// slot inside its aligned group of four is XOR-ed with (tile >> 1) (bank spreading for the readers, see
// qcnn_kernels.h); both tiles of builder wave bw have tile >> 1 == bw, and the wave fetched its code-book rows
// pre-swizzled (mfma_load), so lane group q already holds the rows that belong at position q.  M0 holds the
// full byte address of the tile (measured on gfx950: all of M0 is added, not 16 bits of it); an SALU write of
// M0 needs one wait state before an add-TID LDS instruction reads it (without the s_nop the store uses the
// previous M0: scripts/ubench/addtid_probe.hip).
#define QCNN_WR2(ea, eb, oa, ob)                                                                                   \
  asm volatile("s_mov_b32 m0, %[m]\n\ts_nop 0\n\tds_write_addtid_b32 %[" ea "] offset:%[" oa "]\n\t"              \
               "ds_write_addtid_b32 %[" eb "] offset:%[" ob "]"                                                      \
               :: [m] "s"(m0v), [e0] "v"(v[0]), [e1] "v"(v[1]), [e2] "v"(v[2]), [e3] "v"(v[3]),                      \
                  [o0] "n"(I * 1024), [o1] "n"(I * 1024 + 256), [o2] "n"(I * 1024 + 512), [o3] "n"(I * 1024 + 768)   \
               : "m0", "memory")
template <int I>
__device__ __forceinline__ void store_tile_lo(const f32x4& v, uint32_t m0v) { QCNN_WR2("e0", "e1", "o0", "o1"); }
template <int I>
__device__ __forceinline__ void store_tile_hi(const f32x4& v, uint32_t m0v) { QCNN_WR2("e2", "e3", "o2", "o3"); }
// all four registers of a tile behind ONE M0 write (4-dim first layers: half the scalar instructions of the split form)
template <int I>
__device__ __forceinline__ void store_tile_all(const f32x4& v, uint32_t m0v) {
  asm volatile("s_mov_b32 m0, %[m]\n\ts_nop 0\n\tds_write_addtid_b32 %[e0] offset:%[o0]\n\t"
               "ds_write_addtid_b32 %[e1] offset:%[o1]\n\tds_write_addtid_b32 %[e2] offset:%[o2]\n\t"
               "ds_write_addtid_b32 %[e3] offset:%[o3]"
               :: [m] "s"(m0v), [e0] "v"(v[0]), [e1] "v"(v[1]), [e2] "v"(v[2]), [e3] "v"(v[3]),
                  [o0] "n"(I * 1024), [o1] "n"(I * 1024 + 256), [o2] "n"(I * 1024 + 512), [o3] "n"(I * 1024 + 768)
               : "m0", "memory");
}


struct ConvGeom {
  int W, Cin, knl, M, MG, G, wiL, wiU;
  int slide, hiL, hiU;  // slide: the workgroup sweeps a strip of source rows under a segment of one output column (k_conv_aprx<.., SLIDE>)
  int period;           // slide: slots * stride (the slot -> tap-column map repeats with it)
  uint32_t pixStride;   // bytes from one source pixel to the next: Cin * 512 (panels) or 4 (NCHW input read in place)
  uint32_t rowStride;   // bytes of one (tap, sub-space) row of the assignment table
};
struct StagePos {
  int hi, wi, mg;
  int ph;               // sliding variant: source row modulo the slot period (ConvGeom::period); else unused
};
__device__ __forceinline__ StagePos next_pos(const StagePos& c, const ConvGeom& g) {
  StagePos n = c;
  if (++n.mg == g.MG) {
    n.mg = 0;
    if (++n.wi > g.wiU) {
      n.wi = g.wiL; ++n.hi;
      if (g.slide && ++n.ph == g.period) n.ph = 0;
    }
  }
  return n;
}
__device__ __forceinline__ uint32_t pixel_off(const StagePos& c, const ConvGeom& g) {
  return (uint32_t)(c.hi * g.W + c.wi) * g.pixStride;
}


constexpr uint32_t IDX_LDS = 2u * STAGE_BYTES;     // two row buffers behind the two LUT stages
constexpr uint32_t IDX_BUF = 2048u;
template <int NB>
struct IdxBlk {
  uint32_t w[NB];
};
template <int NB>
__device__ __forceinline__ void blk_load(IdxBlk<NB>& o, const char* __restrict__ src) {
  const uint4* __restrict__ q = reinterpret_cast<const uint4*>(__builtin_assume_aligned(src, 16));
#pragma unroll
  for (int i = 0; i < NB / 4; ++i) {
    const uint4 v = q[i];
    o.w[4 * i] = v.x; o.w[4 * i + 1] = v.y; o.w[4 * i + 2] = v.z; o.w[4 * i + 3] = v.w;
  }
}
// 16 bytes per lane from row (wave-uniform: an SGPR pair) + off (per lane, 32 bits) to LDS byte ldsDst + 16 * lane (ldsDst
// wave-uniform); completion = vmcnt.  The scalar-base form costs no 64-bit vector address arithmetic and no VGPR pair.
__device__ __forceinline__ void glds16(const char* row, uint32_t off, uint32_t ldsDst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(off), "s"(row), "s"(ldsDst) : "memory");
}
template <int BYTES>
__device__ __forceinline__ void idx_row_to_lds(const char* __restrict__ row, uint32_t ldsDst, int lane) {
  static_assert(BYTES <= 3072 && BYTES % 16 == 0, "a workgroup row fits one buffer (2 KB; 3 KB for the fp16-sum tiles of k_conv_sym8)");
  if (lane * 16 < BYTES) glds16(row, (uint32_t)lane * 16u, ldsDst);
  if (BYTES > 1024 && lane * 16 < BYTES - 1024) glds16(row, (uint32_t)lane * 16u + 1024u, ldsDst + 1024u);
  if (BYTES > 2048 && lane * 16 < BYTES - 2048) glds16(row, (uint32_t)lane * 16u + 2048u, ldsDst + 2048u);
}
__device__ __forceinline__ void barrier_after_lds_dma() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 1 when 0 <= d < n, else 0 — in integer arithmetic only: a comparison would make hipcc carry the (wave-uniform)
// result as a lane mask and turn it into the asm blocks' scalar operand through v_cndmask + v_readfirstlane, once per
// position and stage (measured: the validity logic of the four positions of conv1 alone cost 30 % of the layer)
__device__ __forceinline__ int in_range(int d, int n) { return (int)(~(uint32_t)(d | (n - 1 - d)) >> 31); }

// hipFuncAttributeMaxDynamicSharedMemorySize has to be raised once per kernel and DEVICE before the 128 KB launch:
// remember (kernel, device) pairs instead of asking the runtime on every launch.
hipError_t allow_big_lds(const void* kern, int bytes) {
  struct Seen { const void* k; std::atomic<unsigned long long> devMask; };
  static Seen seen[256];
  static std::atomic<int> used{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  const int n = used.load(std::memory_order_acquire);
  for (int i = 0; i < n; ++i)
    if (seen[i].k == kern && (seen[i].devMask.load(std::memory_order_relaxed) & bit)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  for (int i = 0; i < n; ++i)
    if (seen[i].k == kern) { seen[i].devMask.fetch_or(bit); return hipSuccess; }
  const int slot = used.fetch_add(1);
  if (slot < 256) { seen[slot].devMask.store(bit); seen[slot].k = kern; }   // a racing duplicate only costs a repeated call
  return hipSuccess;
}


}  // namespace

#endif  // QCNN_DEV_H_
