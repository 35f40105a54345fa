// qcnn_kernels.hip — hand-written gfx950 (CDNA4) kernels of the Quantized-CNN approximate forward pass.
//
// Hot kernels:
//   k_conv_aprx  fused  GetInPdMat (src/CaffeEva.cc:1261-1296)  +  CalcFeatMap_ConvAprx (:760-868)
//   k_fc_aprx    fused  GetInPdMat                              +  CalcFeatMap_FCntAprx (:968-1025)
// Glue kernels (row a9): ReLU :1027, LRN :1038, max-pool :870, softmax :1098, top-5 :1162,
// NCHW<->panel conversions (:1146-1160, :187-189).
//
// Mapping (see qcnn_kernels.h for the HBM layout, DESIGN.md §3 for the measurements behind it): a lane
// carries an image pair.  A workgroup (16 waves, one per CU) owns one 128-image panel, one tile of output
// positions and one slice of output channels.  The look-up table is never materialised in HBM: it is
// produced one STAGE at a time in LDS — a stage = 128 code-word rows = G = 128/K consecutive sub-spaces of
// one source pixel (conv) or of the input vector (FC) for the 128 images; a row is 512 contiguous bytes
// (+16 B pad) = one conflict-free ds_read_b64 per wave.  Stages are double buffered, one s_barrier per
// stage.  The waves are specialised: four BUILDER waves (one per SIMD) multiply stage s+1 out
// (v_mfma_f32_16x16x4_f32, or ordered VALU mul+add in "exact" mode) and store it, twelve GATHER waves
// keep (positions x channels-per-wave) float2 accumulators in VGPRs and consume stage s.  Stages are
// visited in (pixel row-major, sub-space ascending) order, which for any one output is exactly the
// reference's (kh, kw, m) summation order (:840-863), so with the exact builder conv/FC outputs are
// bit-identical to the reference.  Code-word row indices (uint8) are wave-uniform; they are prefetched
// through the vector memory path and broadcast into SGPRs.
#include "qcnn_kernels.h"

#include <float.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PANEL = QCNN_PANEL;              // images per panel
constexpr int NW = 16;                         // waves per workgroup of the two hot kernels (4 per SIMD)
constexpr int NBW = 4;                         // builder waves (waves 0..3: one per SIMD) — MFMA + LDS writes
constexpr int NGW = NW - NBW;                  // gather waves (waves 4..15) — LDS reads + packed adds
constexpr int ROWB = QCNN_ROW_BYTES;           // LDS bytes per code-word row
constexpr int STAGE_ROWS = QCNN_STAGE_ROWS;
constexpr int STAGE_BYTES = STAGE_ROWS * ROWB;  // 67 584 B; two stages = 132 KB of the 160 KB LDS
constexpr int XROWB = PANEL * 4;               // bytes of one activation row in HBM

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Workgroup barrier WITHOUT the implicit "wait for everything" of __syncthreads(): the builder waves
// wait for their LDS writes only (their operand prefetch of the stage after next stays in flight), the
// gather waves wait for nothing (their look-ups were consumed by the adds; their index prefetch stays
// in flight).  The "memory" clobber keeps the compiler from moving LDS accesses across it.
__device__ __forceinline__ void barrier_after_lds_writes() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void barrier_plain() { asm volatile("s_barrier" ::: "memory"); }

// Role assignment.  The matrix pipe is per SIMD, so the four builder waves must sit on four DIFFERENT
// SIMDs; which SIMD a wave lands on is the dispatcher's choice (not a function of the wave index that
// software may rely on), so every wave publishes its SIMD id (HW_REG_HW_ID[5:4]) through LDS and the
// first wave of each SIMD becomes a builder; the other twelve waves get dense gather indices.  (128 VGPRs
// per wave force exactly four waves per SIMD for a 16-wave workgroup; the smaller instantiations fall back
// to "lowest remaining waves" if a SIMD should have none.)
struct WaveRole {
  bool builder;
  int idx;      // builder: 0..3; gather wave: 0..11
};
__device__ __forceinline__ WaveRole assign_roles(char* lds, int wave, int lane) {
  int* tab = reinterpret_cast<int*>(lds);
  const int simd = (int)(__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)) & 3u);   // hwreg(HW_REG_HW_ID, 4, 2)
  if (lane == 0) tab[wave] = simd;
  __syncthreads();
  // every wave derives the same assignment from the table: first wave of each SIMD, and — should the
  // dispatcher ever have left a SIMD without a wave of this workgroup — the lowest remaining waves
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
// Gather (gather waves).  The code-word ROW INDICES (uint8, 0..127: (m mod G)*K + assignment) of
// CPW consecutive output channels are wave-uniform.  They are prefetched one group ahead as packed
// dwords through the VECTOR memory path (every lane loads the same address; tracked by vmcnt, so the
// prefetch never blocks an lgkmcnt wait of the LDS look-ups) and broadcast into SGPRs with
// v_readfirstlane when the group starts.  Every look-up is then s_bfe_u32 + s_mul (row byte offset),
// v_add_u32 (+ lane*8), ds_read_b64 (image pair), v_pk_add_f32.
// ------------------------------------------------------------------------------------------------
template <int N4>
struct Idx {
  uint32_t w[N4];
};

// per-lane (vector) load of N4 dwords at base + vzero (vzero: a VGPR holding 0 the compiler cannot see through)
template <int N4>
__device__ __forceinline__ void vload_idx(Idx<N4>& o, const uint8_t* __restrict__ ap, uint32_t vzero) {
  const uint32_t* __restrict__ ap4 = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(ap + vzero, 4));
#pragma unroll
  for (int j = 0; j < N4; ++j) o.w[j] = ap4[j];
}
template <int N4>
__device__ __forceinline__ void bcast_idx(Idx<N4>& s, const Idx<N4>& v) {
#pragma unroll
  for (int j = 0; j < N4; ++j) s.w[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.w[j]);
}
// immediate load (fallback for groups that were not prefetched); the result is forced into SGPRs
template <int N4>
__device__ __forceinline__ void sload_idx(Idx<N4>& o, const uint8_t* __restrict__ ap) {
  const uint32_t* __restrict__ ap4 = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(ap, 4));
#pragma unroll
  for (int j = 0; j < N4; ++j) o.w[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)ap4[j]);
}

// One hand-scheduled block of 4 (gather4) or 8 (gather8) look-ups, software-pipelined so that no
// instruction depends on its predecessor: all row indices are extracted on the scalar unit
// (s_and/s_bfe/s_lshr), then all addresses are formed (v_mad_u32_u24: row * 528 + the lane's stage
// address), then all ds_read_b64 are issued back to back, then counted s_waitcnt + v_pk_add_f32 IN PLACE
// (tied operands: an accumulator never changes register).  The counted waits stay correct with other
// lgkm operations outstanding at entry: LDS returns in order, so "at most N outstanding" implies the
// first 8-N reads of the block are back.  `valid` (wave-uniform) = 0 skips the block with a branch
// INSIDE the asm text, so that the compiler sees straight-line code and keeps every accumulator in one
// register for the whole kernel.
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

__device__ __forceinline__ void gather4(f32x2* acc, uint32_t w0, uint32_t base, uint32_t rowb, int valid) {
  f32x2 v0, v1, v2, v3;
  uint32_t a0, a1, a2, a3, t0, t1, t2, t3;
  asm volatile("s_cmp_eq_u32 %[ok], 0\n\ts_cbranch_scc1 .Lqskip%=\n\t"
               QCNN_X4("w0", "t0", "t1", "t2", "t3")
               QCNN_MAD("a0", "t0") QCNN_MAD("a1", "t1") QCNN_MAD("a2", "t2") QCNN_MAD("a3", "t3")
               QCNN_RD("v0", "a0") QCNN_RD("v1", "a1") QCNN_RD("v2", "a2") QCNN_RD("v3", "a3")
               QCNN_ACC("3", "c0", "v0") QCNN_ACC("2", "c1", "v1") QCNN_ACC("1", "c2", "v2") QCNN_ACC("0", "c3", "v3")
               "\n.Lqskip%=:"
               : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [v0] "=&v"(v0),
                 [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2),
                 [a3] "=&v"(a3), [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3)
               : [w0] "s"(w0), [b] "v"(base), [rb] "v"(rowb), [ok] "s"(valid)
               : "scc");
}

// CPW look-ups of one group; `stage` = LDS byte address of the lane's image pair in row 0 of the stage
template <int CPW>
__device__ __forceinline__ void gather_apply(f32x2 (&acc)[CPW], const Idx<CPW / 4>& o, uint32_t stage, int valid) {
  static_assert(CPW % 4 == 0, "indices are fetched as packed dwords");
  uint32_t rowb;
  asm volatile("v_mov_b32 %0, 0x210" : "=v"(rowb));   // QCNN_ROW_BYTES, kept in a VGPR for v_mad_u32_u24
#pragma unroll
  for (int j = 0; j + 1 < CPW / 4; j += 2) gather8(&acc[4 * j], o.w[j], o.w[j + 1], stage, rowb, valid);
  if ((CPW / 4) % 2) gather4(&acc[CPW - 4], o.w[CPW / 4 - 1], stage, rowb, valid);
}

// ------------------------------------------------------------------------------------------------
// LUT stage builders (builder waves).  A stage covers sub-spaces m0 .. m0+G-1 (those < mEnd), K rows
// each.  The 128-image activation row of dim d of sub-space m starts at
// xbase + xoff0 + (m*Cs + d) * 512 bytes.
// ------------------------------------------------------------------------------------------------

// exact: y = ((0 + x0*c0) + x1*c1) + ...  with separately rounded product and sum, the order of the
// reference's saxpy chain (src/CaffeEva.cc:1284-1289, include/BlasWrapper.h:164-184).  Any K <= 128.
// Builder wave bw computes rows bw*ceil(K/4) .. of every sub-space; a lane carries an image pair.
__device__ __forceinline__ void build_stage_exact(char* stage, const char* __restrict__ xbase, uint32_t xoff0,
                                                  const float* __restrict__ ctrd, int K, int Cs, int D, int G, int m0,
                                                  int mEnd, int bw, int lane) {
  const int kpw = (K + NBW - 1) / NBW;
  const int k0 = bw * kpw;
  const int k1 = min(K, k0 + kpw);
  for (int g = 0; g < G; ++g) {
    const int m = m0 + g;
    if (m >= mEnd) break;
    const int dsel = min(D - m * Cs, Cs);
    const char* __restrict__ xm = xbase + xoff0 + (uint32_t)(m * Cs) * (uint32_t)XROWB + lane * 8;
    f32x2 xv[QCNN_MAX_CS];
#pragma unroll
    for (int d = 0; d < QCNN_MAX_CS; ++d) {
      xv[d] = f32x2{0.0f, 0.0f};
      if (d < dsel) xv[d] = *reinterpret_cast<const f32x2*>(xm + d * XROWB);
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
      *reinterpret_cast<f32x2*>(stage + (g * K + k) * ROWB + lane * 8) = f32x2{v0, v1};
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

// Addresses are (wave-uniform pointer) + (32-bit lane offset) so that the loads take the
// "SGPR base + VGPR offset" form and need no per-load 64-bit vector arithmetic.
template <int KT, int KS>
__device__ __forceinline__ void mfma_load(MfmaOps<KT, KS>& o, const char* __restrict__ xbase, uint32_t xoff0,
                                          const float* __restrict__ ctrd, int Cs, int m0, int bw, int lane) {
  constexpr int K = KT * 16;
  constexpr int SUBS = MfmaOps<KT, KS>::SUBS;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t laneA = lk * K + li;                                   // floats
  const uint32_t laneB = lk * XROWB + li * 4;                            // bytes
  const float* __restrict__ cbU = ctrd + (size_t)m0 * Cs * K;            // uniform
  const char* __restrict__ xbU = xbase + xoff0 + (uint32_t)(m0 * Cs) * (uint32_t)XROWB + bw * 128;   // uniform
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int mi = i / KT, kk = (i % KT) * 16;             // compile-time
      o.a[i][ks] = (cbU + ((mi * Cs + ks * 4) * K + kk))[laneA];
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub)
        o.b[it][sub][ks] = *reinterpret_cast<const float*>(xbU + ((sub * Cs + ks * 4) * XROWB + it * 64) + laneB);
  }
}

// Multiply the stage `o` was loaded for (m0, mEnd, D, Cs) out into LDS.  The 16 tiles of the wave are
// walked in pairs with a hand-made software pipeline: MFMA(pair n) is interleaved instruction by
// instruction with the LDS writes of pair n-1, so that a write (which a single wave issues every ~15
// cycles) always sits in the 32-cycle shadow of a matrix instruction and never waits for its own result.
__device__ __forceinline__ f32x4 round_f16(f32x4 v) {
  return f32x4{(float)(_Float16)v[0], (float)(_Float16)v[1], (float)(_Float16)v[2], (float)(_Float16)v[3]};
}

template <int KT, int KS>
__device__ __forceinline__ void mfma_store(MfmaOps<KT, KS>& o, char* stage, int Cs, int D, int m0, int mEnd, int bw,
                                           int lane, int f16) {
  constexpr int SUBS = MfmaOps<KT, KS>::SUBS;
  const int li = lane & 15, lk = lane >> 4;
  // plain: every sub-space of the stage exists and has all 4*KS dims -> nothing to zero
  const bool plain = (m0 + SUBS <= mEnd) && (D - (m0 + SUBS - 1) * Cs >= 4 * KS) && (Cs == 4 * KS);
  if (!plain) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        const bool ok = (m0 + sub < mEnd) && (ks * 4 + lk < min(D - (m0 + sub) * Cs, Cs));
#pragma unroll
        for (int it = 0; it < 2; ++it) o.b[it][sub][ks] = ok ? o.b[it][sub][ks] : 0.0f;
#pragma unroll
        for (int i = sub * KT; i < (sub + 1) * KT; ++i) o.a[i][ks] = ok ? o.a[i][ks] : 0.0f;
      }
    }
  }
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  char* wl = stage + (lk * 4) * ROWB + (bw * 32 + li) * 4;
  f32x4 pa = zero, pb = zero;           // results of the previous pair, still to be written
  char *wa = wl, *wb = wl;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n <= 8; ++n) {        // pair n = tiles (it, i), (it, i+1) with it = n / 4, i = 2 * (n % 4)
    const int it = (n < 8 ? n : 0) / 4, i = 2 * ((n < 8 ? n : 0) % 4);
    f32x4 ca = zero, cb = zero;
    if (n < 8) ca = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][0], o.b[it][i / KT][0], zero, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (n > 0) { *reinterpret_cast<float*>(wa) = pa[0]; *reinterpret_cast<float*>(wa + ROWB) = pa[1]; }
    __builtin_amdgcn_sched_barrier(0);
    if (n < 8) cb = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i + 1][0], o.b[it][(i + 1) / KT][0], zero, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (n > 0) { *reinterpret_cast<float*>(wa + 2 * ROWB) = pa[2]; *reinterpret_cast<float*>(wa + 3 * ROWB) = pa[3]; }
    __builtin_amdgcn_sched_barrier(0);
    if (KS > 1 && n < 8) ca = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][KS - 1], o.b[it][i / KT][KS - 1], ca, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (n > 0) { *reinterpret_cast<float*>(wb) = pb[0]; *reinterpret_cast<float*>(wb + ROWB) = pb[1]; }
    __builtin_amdgcn_sched_barrier(0);
    if (KS > 1 && n < 8)
      cb = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i + 1][KS - 1], o.b[it][(i + 1) / KT][KS - 1], cb, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (n > 0) { *reinterpret_cast<float*>(wb + 2 * ROWB) = pb[2]; *reinterpret_cast<float*>(wb + 3 * ROWB) = pb[3]; }
    __builtin_amdgcn_sched_barrier(0);
    pa = ca; pb = cb;
    if (f16) { pa = round_f16(pa); pb = round_f16(pb); }   // tolerance study only (uniform branch)
    wa = wl + it * 64 + i * 16 * ROWB;
    wb = wa + 16 * ROWB;
  }
}

// ------------------------------------------------------------------------------------------------
// conv.  Workgroup = 16 waves = one 128-image panel x a TH x TW tile of output positions x NC*CPW
// output channels of one group.  Stages (source pixel row-major, sub-space group ascending) are double
// buffered in LDS, one s_barrier per stage.  The waves are SPECIALISED, so that the chain "MFMA -> LDS
// write" of stage s+1 and the chain "LDS read -> add" of stage s run concurrently instead of one after
// the other in every wave:
//   builder waves 0..3 (one per SIMD):  [build stage s+1 from operands in registers] [fetch operands s+2]
//   gather waves 4..15:                 [broadcast indices of stage s] [prefetch indices s+1] [gather stage s]
// The tile is cut into 1 x SW strips; gather wave gw owns strip gw / NC and channels (gw % NC)*CPW ..
// +CPW-1 and keeps SW x CPW float2 accumulators.  KT = K/16 selects the MFMA builder, KT = 0 the exact
// builder (any K <= 128).
// ------------------------------------------------------------------------------------------------
struct ConvGeom {
  int W, Cin, knl, M, Ct, MG, G, wiL, wiU;
};
// Workgroups are dispatched in linear order, so the tiles are numbered heaviest first: interior tiles
// (full receptive field = most stages), then the four edges, then the corners.  With ~5 workgroups per
// CU in the 13x13 layers the last dispatch round is then made of the short border tiles instead of
// whatever row-major order leaves over (longest-processing-time-first).
__device__ __forceinline__ void tile_of_rank(int r, int tilesY, int tilesX, int& ty, int& tx) {
  if (tilesY < 3 || tilesX < 3) { ty = r / tilesX; tx = r % tilesX; return; }
  const int iy = tilesY - 2, ix = tilesX - 2;
  if (r < iy * ix) { ty = 1 + r / ix; tx = 1 + r % ix; return; }
  r -= iy * ix;
  if (r < ix) { ty = 0; tx = 1 + r; return; }
  r -= ix;
  if (r < ix) { ty = tilesY - 1; tx = 1 + r; return; }
  r -= ix;
  if (r < iy) { ty = 1 + r; tx = 0; return; }
  r -= iy;
  if (r < iy) { ty = 1 + r; tx = tilesX - 1; return; }
  r -= iy;
  ty = (r >> 1) ? tilesY - 1 : 0;
  tx = (r & 1) ? tilesX - 1 : 0;
}
struct StagePos {
  int hi, wi, mg;
};
__device__ __forceinline__ StagePos next_pos(const StagePos& c, const ConvGeom& g) {
  StagePos n = c;
  if (++n.mg == g.MG) {
    n.mg = 0;
    if (++n.wi > g.wiU) { n.wi = g.wiL; ++n.hi; }
  }
  return n;
}
__device__ __forceinline__ uint32_t pixel_off(const StagePos& c, const ConvGeom& g) {
  return (uint32_t)(c.hi * g.W + c.wi) * (uint32_t)g.Cin * (uint32_t)XROWB;
}

// indices of the first sub-space of stage c for every position of the wave's strip (taps that do not
// exist are clamped to an existing one: the load is harmless, the gather skips them)
template <int SW, int CPW>
__device__ __forceinline__ void conv_prefetch_idx(Idx<CPW / 4> (&v)[SW], const StagePos& c, const ConvGeom& g,
                                                  const uint8_t* __restrict__ rowsC, int rowStart,
                                                  const int (&colStart)[SW], uint32_t vzero) {
  const int kh = min(max(c.hi - rowStart, 0), g.knl - 1);
#pragma unroll
  for (int dx = 0; dx < SW; ++dx) {
    const int kw = min(max(c.wi - colStart[dx], 0), g.knl - 1);
    vload_idx(v[dx], rowsC + (size_t)((kh * g.knl + kw) * g.M + c.mg * g.G) * g.Ct, vzero);
  }
}

template <int SW, int CPW, bool ONE>
__device__ __forceinline__ void conv_gather(f32x2 (&acc)[SW][CPW], const Idx<CPW / 4> (&first)[SW], const StagePos& c,
                                            const ConvGeom& g, const uint8_t* __restrict__ rowsC, int rowStart,
                                            const int (&colStart)[SW], uint32_t stage) {
  const int kh = c.hi - rowStart;
  const bool rowOk = (unsigned)kh < (unsigned)g.knl;
#pragma unroll
  for (int dx = 0; dx < SW; ++dx) {
    const int kw = c.wi - colStart[dx];
    const int valid = uni((rowOk && (unsigned)kw < (unsigned)g.knl) ? 1 : 0);
    gather_apply<CPW>(acc[dx], first[dx], stage, valid);
    if (!ONE) {                              // further sub-spaces of the stage (K <= 64 only)
      const int m0 = c.mg * g.G;
      const int n = valid ? min(g.M, m0 + g.G) - m0 : 0;
      for (int i = 1; i < n; ++i) {
        Idx<CPW / 4> more;
        sload_idx(more, rowsC + (size_t)((kh * g.knl + kw) * g.M + m0 + i) * g.Ct);
        gather_apply<CPW>(acc[dx], more, stage, 1);
      }
    }
  }
}

template <int TH, int TW, int SW, int CPW, int KT, int KS>
__global__ __launch_bounds__(NW * 64) void k_conv_aprx(ConvParams p, int tilesX, int tilesY, int chunksPerGrp, int G) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  static_assert(TW % SW == 0, "strips tile the row");
  constexpr int SPR = TW / SW;            // strips per tile row
  constexpr int NSTRIP = TH * SPR;
  static_assert(NGW % NSTRIP == 0, "gather waves split evenly over strips");
  constexpr int NC = NGW / NSTRIP;        // channel chunks (of CPW) inside the workgroup
  constexpr int KTT = KT > 0 ? KT : 1;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  int ty, tx;                                       // blockIdx.x = tile rank (heaviest first) * panels + panel
  tile_of_rank((int)(blockIdx.x / (unsigned)p.panels), tilesY, tilesX, ty, tx);
  const int panel = (int)(blockIdx.x % (unsigned)p.panels);
  const int grp = blockIdx.y / chunksPerGrp, chunk = blockIdx.y % chunksPerGrp;
  const int Cg = p.Cin / p.grp, Ctg = p.Ct / p.grp;
  const int M = p.M;

  const int ho0 = ty * TH, wo0 = tx * TW;
  const int hoL = min(ho0 + TH, p.Ho) - 1, woL = min(wo0 + TW, p.Wo) - 1;   // last real position of the tile
  const int hiL = max(0, ho0 * p.stride - p.pad), hiU = min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1);
  ConvGeom g;
  g.W = p.W; g.Cin = p.Cin; g.knl = p.knl; g.M = M; g.Ct = p.Ct; g.G = G;
  g.MG = (M + G - 1) / G;                           // stages per source pixel
  g.wiL = max(0, wo0 * p.stride - p.pad);
  g.wiU = min(p.W - 1, woL * p.stride - p.pad + p.knl - 1);
  const int S = (hiU - hiL + 1) * (g.wiU - g.wiL + 1) * g.MG;
  const int Sp = (S + 1) & ~1;                      // every wave runs Sp stage periods (barriers)
  const StagePos first = {hiL, g.wiL, 0};

  const WaveRole role = assign_roles(lds, wave, lane);
  if (role.builder) {
    // ---------------------------------------------------------------- builder wave ----
    const int bw = role.idx;
    const int K = p.K, Cs = p.Cs;
    const char* __restrict__ xbase =
        reinterpret_cast<const char*>(p.src + ((size_t)panel * p.H * p.W * p.Cin + (size_t)grp * Cg) * PANEL);
    // Two operand sets: while stage s+1 is multiplied out of one, the other one already holds (or is
    // receiving) stage s+2, and the loads of stage s+3 are issued as soon as the first is consumed, so
    // an operand fetch has a whole stage period to land.
    MfmaOps<KTT, KS> opsA, opsB;
    StagePos q1 = next_pos(first, g);
    StagePos q2 = next_pos(q1, g);
    StagePos q3 = next_pos(q2, g);
    if (KT > 0) {
      mfma_load<KTT, KS>(opsA, xbase, pixel_off(first, g), p.ctrd, Cs, 0, bw, lane);
      mfma_store<KTT, KS>(opsA, lds, Cs, Cg, 0, M, bw, lane, p.lutF16);
      {
        const StagePos qa = (q1.hi > hiU) ? first : q1, qb = (q2.hi > hiU) ? first : q2;
        mfma_load<KTT, KS>(opsA, xbase, pixel_off(qa, g), p.ctrd, Cs, qa.mg * G, bw, lane);
        __builtin_amdgcn_sched_barrier(0);             // keep set A's loads older than set B's (vmcnt accounting)
        mfma_load<KTT, KS>(opsB, xbase, pixel_off(qb, g), p.ctrd, Cs, qb.mg * G, bw, lane);
      }
    } else {
      build_stage_exact(lds, xbase, pixel_off(first, g), p.ctrd, K, Cs, Cg, G, 0, M, bw, lane);
    }
    barrier_after_lds_writes();
    // Straight-line body (no VMEM operation under a condition), so that the compiler's vmcnt waits are
    // exact: "all but the 2*(8+2*SUBS) loads of the other set".  Sp rounds S up to even; the surplus
    // stage is built from re-fetched operands into a buffer nobody reads.
    for (int s = 0; s < Sp; s += 2) {
      if (KT > 0) {                                    // stage s+1 -> buffer 1, from set A
        mfma_store<KTT, KS>(opsA, lds + STAGE_BYTES, Cs, Cg, q1.mg * G, M, bw, lane, p.lutF16);
        const StagePos qf = (q3.hi > hiU) ? first : q3;
        mfma_load<KTT, KS>(opsA, xbase, pixel_off(qf, g), p.ctrd, Cs, qf.mg * G, bw, lane);
      } else if (s + 1 < S) {
        build_stage_exact(lds + STAGE_BYTES, xbase, pixel_off(q1, g), p.ctrd, K, Cs, Cg, G, q1.mg * G, M, bw, lane);
      }
      barrier_after_lds_writes();
      q1 = q2; q2 = q3; q3 = next_pos(q3, g);
      if (KT > 0) {                                    // stage s+2 -> buffer 0, from set B
        mfma_store<KTT, KS>(opsB, lds, Cs, Cg, q1.mg * G, M, bw, lane, p.lutF16);
        const StagePos qf = (q3.hi > hiU) ? first : q3;
        mfma_load<KTT, KS>(opsB, xbase, pixel_off(qf, g), p.ctrd, Cs, qf.mg * G, bw, lane);
      } else if (s + 2 < S) {
        build_stage_exact(lds, xbase, pixel_off(q1, g), p.ctrd, K, Cs, Cg, G, q1.mg * G, M, bw, lane);
      }
      barrier_after_lds_writes();
      q1 = q2; q2 = q3; q3 = next_pos(q3, g);
    }
    return;
  }

  // ------------------------------------------------------------------ gather wave ----
  const int gw = role.idx;
  const int strip = gw / NC, cc = gw % NC;
  const int sdy = strip / SPR, sdx0 = (strip % SPR) * SW;
  const int cw0 = chunk * (NC * CPW) + cc * CPW;     // first channel of this wave inside the group
  const int ccnt = min(CPW, Ctg - cw0);
  const int ho = ho0 + sdy;
  const bool active = ccnt > 0 && ho < p.Ho;         // waves without channels / outside the map only keep the barriers
  const int c0 = grp * Ctg + (active ? cw0 : 0);
  const uint8_t* __restrict__ rowsC = p.rows + c0;
  uint32_t vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const uint32_t ldsBase = (uint32_t)(uintptr_t)lds;   // LDS byte address of the dynamic segment

  f32x2 acc[SW][CPW];
  {
    const float* __restrict__ bp = p.bias + c0;   // reads past the last channel stay inside the arena
#pragma unroll
    for (int cb = 0; cb < CPW; cb += 4) {         // four at a time: few bias temporaries alive
      float b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = bp[cb + j];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int dx = 0; dx < SW; ++dx) acc[dx][cb + j] = f32x2{b[j], b[j]};
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // first source row / column of the strip's positions; positions outside the map get a start that can
  // never match a tap
  const int rowStart = active ? ho * p.stride - p.pad : -(1 << 28);
  int colStart[SW];
#pragma unroll
  for (int dx = 0; dx < SW; ++dx) {
    const int wo = wo0 + sdx0 + dx;
    colStart[dx] = (wo < p.Wo) ? wo * p.stride - p.pad : -(1 << 28);
  }

  // Per stage: [gather stage s][broadcast the indices of stage s+1][prefetch those of stage s+2][barrier].
  // The index hand-over sits BEFORE the barrier, i.e. in the time an early wave would spend waiting for
  // the slowest one anyway; right after the barrier every gather wave starts reading LDS.
  Idx<CPW / 4> vidx[SW], sidx[SW];
  StagePos c0p = first;
  StagePos c1p = next_pos(c0p, g);
  conv_prefetch_idx<SW, CPW>(vidx, c0p, g, rowsC, rowStart, colStart, vzero);
#pragma unroll
  for (int dx = 0; dx < SW; ++dx) bcast_idx(sidx[dx], vidx[dx]);
  conv_prefetch_idx<SW, CPW>(vidx, c1p, g, rowsC, rowStart, colStart, vzero);
  __builtin_amdgcn_s_setprio(2);   // the gather waves are the critical path of a stage: they win issue arbitration
                                   // against the builder of their SIMD (+1.6 %; a priority rising with the wave age
                                   // to equalise arrival at the barrier measured the same)
  barrier_plain();
  for (int s = 0; s < Sp; ++s) {
    conv_gather<SW, CPW, KT == 8>(acc, sidx, c0p, g, rowsC, (s < S) ? rowStart : -(1 << 28), colStart,
                                  ldsBase + (uint32_t)((s & 1) * STAGE_BYTES + lane * 8));
    c0p = c1p; c1p = next_pos(c1p, g);
#pragma unroll
    for (int dx = 0; dx < SW; ++dx) bcast_idx(sidx[dx], vidx[dx]);                   // indices of stage s+1
    conv_prefetch_idx<SW, CPW>(vidx, c1p, g, rowsC, rowStart, colStart, vzero);    // of stage s+2 (past the end: clamped, unused)
    barrier_plain();
  }

  if (active) {
    float* __restrict__ dst = p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL;
#pragma unroll
    for (int dx = 0; dx < SW; ++dx) {
      const int wo = wo0 + sdx0 + dx;
      if (wo < p.Wo) {
        float* o = dst + ((size_t)(ho * p.Wo + wo) * p.Ct + c0) * PANEL + 2 * lane;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          if (c < ccnt) {
            f32x2 v = acc[dx][c];
            if (p.relu) {
              v.x = (0.0f < v.x) ? v.x : 0.0f;
              v.y = (0.0f < v.y) ? v.y : 0.0f;
            }
            *reinterpret_cast<f32x2*>(o + c * PANEL) = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fully connected: stages of G sub-spaces; 4 builder waves + 12 gather waves x CPW channels, as in the
// conv kernel; the gather waves walk a stream of groups (one sub-space each) with the indices of the
// next group always in flight.  Optional split over the sub-space axis (blockIdx.z): partial sums go
// to p.partial and are reduced by k_sum_partials.
// ------------------------------------------------------------------------------------------------
template <int CPW, int KT, int KS>
__global__ __launch_bounds__(NW * 64) void k_fc_aprx(FcParams p, int G, int stagesPerSplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int KTT = KT > 0 ? KT : 1;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int panel = blockIdx.y;
  const int split = blockIdx.z;
  const int M = p.M;
  const int mBeg = split * stagesPerSplit * G;
  const int mEnd = min(M, mBeg + stagesPerSplit * G);
  const int S = (mEnd - mBeg + G - 1) / G;
  const int Sp = (S + 1) & ~1;                      // every wave runs Sp stage periods (barriers)

  const WaveRole role = assign_roles(lds, wave, lane);
  if (role.builder) {
    const int bw = role.idx;
    const int K = p.K, Cs = p.Cs;
    const char* __restrict__ xbase = reinterpret_cast<const char*>(p.src + (size_t)panel * p.D * PANEL);
    MfmaOps<KTT, KS> opsA, opsB;                          // two operand sets, see k_conv_aprx
    const int mLastStage = mBeg + max(S - 1, 0) * G;  // operand prefetches past the end re-fetch the last stage
    if (S > 0) {
      if (KT > 0) {
        mfma_load<KTT, KS>(opsA, xbase, 0u, p.ctrd, Cs, mBeg, bw, lane);
        mfma_store<KTT, KS>(opsA, lds, Cs, p.D, mBeg, mEnd, bw, lane, p.lutF16);
        mfma_load<KTT, KS>(opsA, xbase, 0u, p.ctrd, Cs, min(mBeg + G, mLastStage), bw, lane);
        __builtin_amdgcn_sched_barrier(0);
        mfma_load<KTT, KS>(opsB, xbase, 0u, p.ctrd, Cs, min(mBeg + 2 * G, mLastStage), bw, lane);
      } else {
        build_stage_exact(lds, xbase, 0u, p.ctrd, K, Cs, p.D, G, mBeg, mEnd, bw, lane);
      }
    }
    barrier_after_lds_writes();
    for (int s = 0; s < Sp; s += 2) {                  // straight-line body, see k_conv_aprx
      const int m0 = mBeg + s * G;
      if (KT > 0) {
        mfma_store<KTT, KS>(opsA, lds + STAGE_BYTES, Cs, p.D, min(m0 + G, mLastStage), mEnd, bw, lane, p.lutF16);
        mfma_load<KTT, KS>(opsA, xbase, 0u, p.ctrd, Cs, min(m0 + 3 * G, mLastStage), bw, lane);
      } else if (s + 1 < S) {
        build_stage_exact(lds + STAGE_BYTES, xbase, 0u, p.ctrd, K, Cs, p.D, G, m0 + G, mEnd, bw, lane);
      }
      barrier_after_lds_writes();
      if (KT > 0) {
        mfma_store<KTT, KS>(opsB, lds, Cs, p.D, min(m0 + 2 * G, mLastStage), mEnd, bw, lane, p.lutF16);
        mfma_load<KTT, KS>(opsB, xbase, 0u, p.ctrd, Cs, min(m0 + 4 * G, mLastStage), bw, lane);
      } else if (s + 2 < S) {
        build_stage_exact(lds, xbase, 0u, p.ctrd, K, Cs, p.D, G, m0 + 2 * G, mEnd, bw, lane);
      }
      barrier_after_lds_writes();
    }
    return;
  }

  const int gw = role.idx;
  const int cw0r = blockIdx.x * (NGW * CPW) + gw * CPW;
  const int ccnt = min(CPW, p.Ct - cw0r);
  const bool active = ccnt > 0;
  const int cw0 = active ? cw0r : 0;
  const uint8_t* __restrict__ rowsC = p.rows + cw0;
  uint32_t vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const uint32_t ldsBase = (uint32_t)(uintptr_t)lds;   // LDS byte address of the dynamic segment

  f32x2 acc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) acc[c] = f32x2{0.0f, 0.0f};
  if (split == 0) {
    const float* __restrict__ bp = p.bias + cw0;   // over-read stays inside the arena
#pragma unroll
    for (int cb = 0; cb < CPW; cb += 4) {
      float b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = bp[cb + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[cb + j] = f32x2{b[j], b[j]};
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // group stream: sidx always holds the (already broadcast) indices of the next group to gather, vidx
  // the prefetch of the one after it, so that the hand-over never sits right behind a barrier
  Idx<CPW / 4> vidx, sidx;
  const int mClamp = max(mEnd - 1, mBeg);
  vload_idx(vidx, rowsC + (size_t)mBeg * p.Ct, vzero);
  bcast_idx(sidx, vidx);
  vload_idx(vidx, rowsC + (size_t)min(mBeg + 1, mClamp) * p.Ct, vzero);
  __builtin_amdgcn_s_setprio(2);   // see k_conv_aprx
  barrier_plain();
  for (int s = 0; s < Sp; ++s) {
    const int m0 = mBeg + s * G;
    const int mLast = min(mEnd, m0 + G);
    const uint32_t stage = ldsBase + (uint32_t)((s & 1) * STAGE_BYTES + lane * 8);
    if (active) {
      for (int m = m0; m < mLast; ++m) {
        gather_apply<CPW>(acc, sidx, stage, 1);
        bcast_idx(sidx, vidx);
        vload_idx(vidx, rowsC + (size_t)min(m + 2, mClamp) * p.Ct, vzero);
      }
    }
    barrier_plain();
  }

  if (active) {
    float* base = (p.msplit > 1) ? p.partial + (size_t)split * p.panels * p.Ct * PANEL : p.dst;
    float* o = base + ((size_t)panel * p.Ct + cw0) * PANEL + 2 * lane;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (c < ccnt) {
        f32x2 v = acc[c];
        if (p.relu && p.msplit == 1) {
          v.x = (0.0f < v.x) ? v.x : 0.0f;
          v.y = (0.0f < v.y) ? v.y : 0.0f;
        }
        *reinterpret_cast<f32x2*>(o + c * PANEL) = v;
      }
    }
  }
}

// dst row e = src row map[e] (the NHWC -> NCHW flatten in front of the first FC layer, src/CaffeEva.cc:187-189)
__global__ void k_permute_rows(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ map,
                               int D, int panels) {
  const int lane = threadIdx.x & 63;
  const size_t rows = (size_t)panels * D;
  for (size_t r = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6); r < rows;
       r += (size_t)gridDim.x * (blockDim.x >> 6)) {
    const size_t panel = r / D;
    const int e = (int)(r % D);
    *reinterpret_cast<f32x2*>(dst + r * PANEL + 2 * lane) =
        *reinterpret_cast<const f32x2*>(src + (panel * D + map[e]) * PANEL + 2 * lane);
  }
}

// dst = partial[0] + partial[1] + ... (fixed order), optional ReLU
__global__ void k_sum_partials(const float4* __restrict__ partial, float4* __restrict__ dst, int msplit, size_t n4,
                               int relu) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = partial[i];
    for (int z = 1; z < msplit; ++z) {
      const float4 w = partial[(size_t)z * n4 + i];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if (relu) {
      v.x = (0.0f < v.x) ? v.x : 0.0f;
      v.y = (0.0f < v.y) ? v.y : 0.0f;
      v.z = (0.0f < v.z) ? v.z : 0.0f;
      v.w = (0.0f < v.w) ? v.w : 0.0f;
    }
    dst[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// glue kernels (rows of 128 images; a lane handles an image pair)
// ------------------------------------------------------------------------------------------------
__global__ void k_relu(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = src[i];
    v.x = (0.0f < v.x) ? v.x : 0.0f;
    v.y = (0.0f < v.y) ? v.y : 0.0f;
    v.z = (0.0f < v.z) ? v.z : 0.0f;
    v.w = (0.0f < v.w) ? v.w : 0.0f;
    dst[i] = v;
  }
}

// LRN, streaming form.  A thread owns one pixel and four images (float4: 32 lanes = one 512-byte row, a
// wave = two pixels) and walks the channels once, keeping the window of N scaled squares and raw values
// in registers: every element is read exactly once.  Same arithmetic and the same summation order as
// k_lrn below (window j ascending; channels outside [0, C) contribute an exact +0.0f instead of being
// skipped, which leaves s > 0 bit-identical).
template <int N>
__global__ __launch_bounds__(256) void k_lrn_stream(const float4* __restrict__ src, float4* __restrict__ dst,
                                                    size_t pixels, int C, float coeff, float nbet, float ini) {
  constexpr int RAD = (N - 1) / 2;
  const size_t px = blockIdx.x * (size_t)(blockDim.x >> 5) + (threadIdx.x >> 5);
  if (px >= pixels) return;
  const int q = threadIdx.x & 31;
  const float4* __restrict__ x = src + px * (size_t)C * 32 + q;
  float4* __restrict__ y = dst + px * (size_t)C * 32 + q;
  const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4 raw[N], sq[N];          // ring: slot (t % N) holds channel t
#pragma unroll
  for (int j = 0; j < N; ++j) { raw[j] = zero; sq[j] = zero; }
  // channels 0 .. RAD-1 enter the window before the first output
#pragma unroll
  for (int t = 0; t < RAD; ++t) {
    const float4 v = (t < C) ? x[(size_t)t * 32] : zero;
    raw[t % N] = v;
    sq[t % N] = make_float4(__fmul_rn(__fmul_rn(v.x, v.x), coeff), __fmul_rn(__fmul_rn(v.y, v.y), coeff),
                            __fmul_rn(__fmul_rn(v.z, v.z), coeff), __fmul_rn(__fmul_rn(v.w, v.w), coeff));
  }
  for (int c0 = 0; c0 < C; c0 += N) {
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const int c = c0 + u;                 // output channel; slot of channel k is (k + N*8) % N = (u + k - c) % N
      const int tin = c + RAD;              // channel entering the window
      const float4 v = (tin < C) ? x[(size_t)tin * 32] : zero;
      raw[(u + RAD) % N] = v;
      sq[(u + RAD) % N] = make_float4(__fmul_rn(__fmul_rn(v.x, v.x), coeff), __fmul_rn(__fmul_rn(v.y, v.y), coeff),
                                      __fmul_rn(__fmul_rn(v.z, v.z), coeff), __fmul_rn(__fmul_rn(v.w, v.w), coeff));
      if (c < C) {
        float4 sacc = make_float4(ini, ini, ini, ini);
#pragma unroll
        for (int j = 0; j < N; ++j) {       // window channel c - RAD + j lives in slot (u - RAD + j) mod N
          const float4 w = sq[(u - RAD + j + N) % N];
          sacc.x = __fadd_rn(sacc.x, w.x); sacc.y = __fadd_rn(sacc.y, w.y);
          sacc.z = __fadd_rn(sacc.z, w.z); sacc.w = __fadd_rn(sacc.w, w.w);
        }
        const float4 xc = raw[u % N];
        float4 o;
        o.x = __fmul_rn(xc.x, expf(__fmul_rn(nbet, logf(sacc.x))));
        o.y = __fmul_rn(xc.y, expf(__fmul_rn(nbet, logf(sacc.y))));
        o.z = __fmul_rn(xc.z, expf(__fmul_rn(nbet, logf(sacc.z))));
        o.w = __fmul_rn(xc.w, expf(__fmul_rn(nbet, logf(sacc.w))));
        y[(size_t)c * 32] = o;
      }
    }
  }
}

// src/CaffeEva.cc:1038-1089: s = k; s += (x*x)*(alpha/n) over the channel window, j ascending (zero pad);
// y = x * expf(-beta * logf(s)).
__global__ void k_lrn(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int C, int lrnSiz,
                      float coeff, float nbet, float ini) {
  const int lane = threadIdx.x & 63;
  const int rad = (lrnSiz - 1) / 2;
  for (size_t r = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6); r < rows;
       r += (size_t)gridDim.x * (blockDim.x >> 6)) {
    const int c = (int)(r % C);
    const float* x = src + r * PANEL + 2 * lane;
    float s0 = ini, s1 = ini;
    for (int j = 0; j < lrnSiz; ++j) {
      const int cc = c - rad + j;
      if (cc >= 0 && cc < C) {
        const f32x2 xv = *reinterpret_cast<const f32x2*>(x + (ptrdiff_t)(cc - c) * PANEL);
        s0 = __fadd_rn(s0, __fmul_rn(__fmul_rn(xv.x, xv.x), coeff));
        s1 = __fadd_rn(s1, __fmul_rn(__fmul_rn(xv.y, xv.y), coeff));
      }
    }
    const f32x2 xc = *reinterpret_cast<const f32x2*>(x);
    f32x2 y;
    y.x = __fmul_rn(xc.x, expf(__fmul_rn(nbet, logf(s0))));
    y.y = __fmul_rn(xc.y, expf(__fmul_rn(nbet, logf(s1))));
    *reinterpret_cast<f32x2*>(dst + r * PANEL + 2 * lane) = y;
  }
}

// src/CaffeEva.cc:870-921: ceil-mode grid, window clipped to the image, std::max(src, dst)
__global__ void k_pool(const float* __restrict__ src, float* __restrict__ dst, int panels, int H, int W, int C,
                       int Ho, int Wo, int knl, int stride, int pad) {
  const int lane = threadIdx.x & 63;
  const size_t rows = (size_t)panels * Ho * Wo * C;
  for (size_t r = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6); r < rows;
       r += (size_t)gridDim.x * (blockDim.x >> 6)) {
    const int c = (int)(r % C);
    size_t q = r / C;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho);
    const int panel = (int)(q / Ho);
    const int hL = max(0, ho * stride - pad), hU = min(H, ho * stride + knl - pad) - 1;
    const int wL = max(0, wo * stride - pad), wU = min(W, wo * stride + knl - pad) - 1;
    const float* base = src + (size_t)panel * H * W * C * PANEL + 2 * lane;
    f32x2 v = {0.0f, 0.0f};
    bool first = true;
    for (int h = hL; h <= hU; ++h)
      for (int w = wL; w <= wU; ++w) {
        const f32x2 s = *reinterpret_cast<const f32x2*>(base + ((size_t)(h * W + w) * C + c) * PANEL);
        if (first) {
          v = s;
        } else {
          v.x = (s.x < v.x) ? v.x : s.x;
          v.y = (s.y < v.y) ? v.y : s.y;
        }
        first = false;
      }
    *reinterpret_cast<f32x2*>(dst + r * PANEL + 2 * lane) = v;
  }
}

// max-pool, four images per thread (32 lanes = one row, a wave = two adjacent channels): same window rule
__global__ __launch_bounds__(256) void k_pool4(const float4* __restrict__ src, float4* __restrict__ dst, int panels,
                                               int H, int W, int C, int Ho, int Wo, int knl, int stride, int pad) {
  const size_t rows = (size_t)panels * Ho * Wo * C;
  const size_t r = blockIdx.x * (size_t)(blockDim.x >> 5) + (threadIdx.x >> 5);
  if (r >= rows) return;
  const int q = threadIdx.x & 31;
  const int c = (int)(r % C);
  size_t t = r / C;
  const int wo = (int)(t % Wo);
  t /= Wo;
  const int ho = (int)(t % Ho);
  const int panel = (int)(t / Ho);
  const int hL = max(0, ho * stride - pad), hU = min(H, ho * stride + knl - pad) - 1;
  const int wL = max(0, wo * stride - pad), wU = min(W, wo * stride + knl - pad) - 1;
  const float4* base = src + (size_t)panel * H * W * C * 32 + q;
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  bool first = true;
  for (int h = hL; h <= hU; ++h)
    for (int w = wL; w <= wU; ++w) {
      const float4 sv = base[((size_t)(h * W + w) * C + c) * 32];
      if (first) {
        v = sv;
      } else {
        v.x = (sv.x < v.x) ? v.x : sv.x;
        v.y = (sv.y < v.y) ? v.y : sv.y;
        v.z = (sv.z < v.z) ? v.z : sv.z;
        v.w = (sv.w < v.w) ? v.w : sv.w;
      }
      first = false;
    }
  dst[r * 32 + q] = v;
}

// Softmax through LDS: a block = 32 images x 8 class lanes.  expf of every logit in parallel into an
// LDS tile [C][32], the reference's SEQUENTIAL float sum over the classes (src/CaffeEva.cc:1107-1114) by
// one thread per image out of LDS, then the division in parallel.  Same values as k_softmax.
__global__ __launch_bounds__(256) void k_softmax_lds(const float* __restrict__ src, float* __restrict__ dst, int C) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* tile = reinterpret_cast<float*>(lds);            // [C][32]
  float* sums = tile + (size_t)C * 32;                    // [32]
  const int img = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const size_t off = (size_t)blockIdx.y * C * PANEL + blockIdx.x * 32 + img;
  const float* x = src + off;
  float* y = dst + off;
  for (int c = cl; c < C; c += 8) tile[c * 32 + img] = expf(x[(size_t)c * PANEL]);
  __syncthreads();
  if (cl == 0) {
    float sum = 0.0f;
    for (int c = 0; c < C; ++c) sum = __fadd_rn(sum, tile[c * 32 + img]);
    sums[img] = sum;
  }
  __syncthreads();
  const float sum = sums[img];
  for (int c = cl; c < C; c += 8) y[(size_t)c * PANEL] = __fdiv_rn(tile[c * 32 + img], sum);
}

// Top-5 through LDS: a block = 32 images x 8 class lanes; per sweep every class lane finds the first
// maximum (strict '<' from FLT_MIN) of its classes, the eight candidates are merged (larger value, then
// lower index = the sequential sweep's first occurrence), the winner is zeroed (src/CaffeEva.cc:1173-1188).
__global__ __launch_bounds__(256) void k_top5_lds(const float* __restrict__ prob, uint16_t* __restrict__ out, int n,
                                                  int C) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* tile = reinterpret_cast<float*>(lds);            // [C][32]
  float* candV = tile + (size_t)C * 32;                   // [8][32]
  int* candI = reinterpret_cast<int*>(candV + 8 * 32);    // [8][32]
  const int img = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int gi = blockIdx.y * PANEL + blockIdx.x * 32 + img;
  const float* x = prob + (size_t)blockIdx.y * C * PANEL + blockIdx.x * 32 + img;
  for (int c = cl; c < C; c += 8) tile[c * 32 + img] = x[(size_t)c * PANEL];
  __syncthreads();
  for (int r = 0; r < 5; ++r) {
    float best = FLT_MIN;
    int bi = 0;
    for (int c = cl; c < C; c += 8) {
      const float v = tile[c * 32 + img];
      if (best < v) { best = v; bi = c; }
    }
    candV[cl * 32 + img] = best;
    candI[cl * 32 + img] = bi;
    __syncthreads();
    if (cl == 0) {
      float b = candV[img];
      int i = candI[img];
      for (int k = 1; k < 8; ++k) {
        const float v = candV[k * 32 + img];
        const int vi = candI[k * 32 + img];
        if (b < v || (b == v && vi < i)) { b = v; i = vi; }
      }
      if (gi < n) out[(size_t)gi * 5 + r] = (uint16_t)i;
      tile[i * 32 + img] = 0.0f;
    }
    __syncthreads();
  }
}

// src/CaffeEva.cc:1098-1116: y = expf(x); sequential float sum over classes; y /= sum.  One thread = one image.
__global__ void k_softmax(const float* __restrict__ src, float* __restrict__ dst, int panels, int C) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= panels * PANEL) return;
  const int panel = t / PANEL, img = t % PANEL;
  const float* x = src + (size_t)panel * C * PANEL + img;
  float* y = dst + (size_t)panel * C * PANEL + img;
  float sum = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float e = expf(x[(size_t)c * PANEL]);
    y[(size_t)c * PANEL] = e;
    sum = __fadd_rn(sum, e);
  }
  for (int c = 0; c < C; ++c) y[(size_t)c * PANEL] = __fdiv_rn(y[(size_t)c * PANEL], sum);
}

// src/CaffeEva.cc:1173-1188: five arg-max sweeps, strict '<' from FLT_MIN, winner zeroed, lowest index wins.
__global__ void k_top5(const float* __restrict__ prob, uint16_t* __restrict__ out, int n, int C) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= n) return;
  const float* x = prob + (size_t)(img / PANEL) * C * PANEL + (img % PANEL);
  int picked[5];
  for (int r = 0; r < 5; ++r) {
    float best = FLT_MIN;
    int bi = 0;
    for (int c = 0; c < C; ++c) {
      float v = x[(size_t)c * PANEL];
      for (int q = 0; q < r; ++q)
        if (picked[q] == c) v = 0.0f;
      if (best < v) {
        best = v;
        bi = c;
      }
    }
    picked[r] = bi;
    out[(size_t)img * 5 + r] = (uint16_t)bi;
  }
}

// [n][E] rows -> panels [E][128] through a 128 x 64 LDS tile (both sides coalesced).
// NCHW: input element e = (c*H + h)*W + w of an image lands in row (h*W + w)*C + c (src/CaffeEva.cc:1146-1160).
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ in, float* __restrict__ dst, int n, int E,
                                              int C, int HW, int nchw) {
  __shared__ float tile[PANEL][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e0 = blockIdx.x * 64;
  const int panel = blockIdx.y;
  for (int i = wave; i < PANEL; i += 4) {
    const int img = panel * PANEL + i;
    const int e = e0 + lane;
    tile[i][lane] = (img < n && e < E) ? in[(size_t)img * E + e] : 0.0f;
  }
  __syncthreads();
  for (int j = wave; j < 64; j += 4) {
    const int e = e0 + j;
    if (e < E) {
      int row = e;
      if (nchw) {
        const int c = e / HW, hw = e % HW;
        row = hw * C + c;
      }
      *reinterpret_cast<f32x2*>(dst + ((size_t)panel * E + row) * PANEL + 2 * lane) =
          f32x2{tile[2 * lane][j], tile[2 * lane + 1][j]};
    }
  }
}

// Device-side input pipeline: 8-bit planar images [n][C][Hs][Ws] (the B, G, R planes as
// BmpImgIO::LoadBmpImg stores them, src/BmpImgIO.cc:84-96) minus the mean image [C][Hs][Ws]
// (RmMeanImg, :203-224), centre crop to H x W (CropImg, :180-201), straight into the panel layout.  Same
// arithmetic as the host path — float(pixel) - mean — so the result is bit-identical to packing the
// host-preprocessed fp32 image, at a quarter of the PCIe bytes.
__global__ __launch_bounds__(256) void k_pack_u8(const uint8_t* __restrict__ in, const float* __restrict__ mean,
                                                 float* __restrict__ dst, int n, int C, int H, int W, int Hs, int Ws) {
  __shared__ float tile[PANEL][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int HW = H * W, E = C * HW;
  const int oy = (Hs - H) / 2, ox = (Ws - W) / 2;
  const int e0 = blockIdx.x * 64;
  const int panel = blockIdx.y;
  const int e = e0 + lane;
  size_t soff = 0;
  float m = 0.0f;
  if (e < E) {
    const int c = e / HW, y = (e % HW) / W, x = e % W;
    soff = ((size_t)c * Hs + (y + oy)) * Ws + (x + ox);
    if (mean) m = mean[soff];
  }
  const size_t srcImg = (size_t)C * Hs * Ws;
  for (int i = wave; i < PANEL; i += 4) {
    const int img = panel * PANEL + i;
    tile[i][lane] = (img < n && e < E) ? ((float)in[(size_t)img * srcImg + soff] - m) : 0.0f;
  }
  __syncthreads();
  for (int j = wave; j < 64; j += 4) {
    const int ee = e0 + j;
    if (ee < E) {
      const int c = ee / HW, hw = ee % HW;
      *reinterpret_cast<f32x2*>(dst + ((size_t)panel * E + (size_t)hw * C + c) * PANEL + 2 * lane) =
          f32x2{tile[2 * lane][j], tile[2 * lane + 1][j]};
    }
  }
}

// panels [E][128] -> [n][E]
__global__ __launch_bounds__(256) void k_unpack(const float* __restrict__ src, float* __restrict__ out, int n, int E) {
  __shared__ float tile[64][PANEL + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e0 = blockIdx.x * 64;
  const int panel = blockIdx.y;
  for (int j = wave; j < 64; j += 4) {
    const int e = e0 + j;
    f32x2 v = {0.0f, 0.0f};
    if (e < E) v = *reinterpret_cast<const f32x2*>(src + ((size_t)panel * E + e) * PANEL + 2 * lane);
    tile[j][2 * lane] = v.x;
    tile[j][2 * lane + 1] = v.y;
  }
  __syncthreads();
  for (int i = wave; i < PANEL; i += 4) {
    const int img = panel * PANEL + i;
    const int e = e0 + lane;
    if (img < n && e < E) out[(size_t)img * E + e] = tile[lane][i];
  }
}

inline int panels_of(int n) { return (n + PANEL - 1) / PANEL; }

template <int TH, int TW, int SW, int CPW>
hipError_t launch_conv(const ConvParams& p, int lutMode, hipStream_t st) {
  constexpr int NC = NGW / (TH * (TW / SW));
  const int Ctg = p.Ct / p.grp;
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH;
  const int chunksPerGrp = (Ctg + NC * CPW - 1) / (NC * CPW);
  const dim3 grid(tilesX * tilesY * p.panels, chunksPerGrp * p.grp, 1);
  const size_t shm = (size_t)2 * STAGE_BYTES;
  const int G = qcnn_stage_group(p.K);
  const bool two = min(p.Cin / p.grp, p.Cs) > 4;      // MFMA k-steps (4 dims each) that carry data
  auto kern = k_conv_aprx<TH, TW, SW, CPW, 0, 1>;
  if (lutMode == 1 && p.K == 128) kern = two ? k_conv_aprx<TH, TW, SW, CPW, 8, 2> : k_conv_aprx<TH, TW, SW, CPW, 8, 1>;
  if (lutMode == 1 && p.K == 64) kern = two ? k_conv_aprx<TH, TW, SW, CPW, 4, 2> : k_conv_aprx<TH, TW, SW, CPW, 4, 1>;
  if (lutMode == 1 && p.K == 32) kern = two ? k_conv_aprx<TH, TW, SW, CPW, 2, 2> : k_conv_aprx<TH, TW, SW, CPW, 2, 1>;
  if (lutMode == 1 && p.K == 16) kern = two ? k_conv_aprx<TH, TW, SW, CPW, 1, 2> : k_conv_aprx<TH, TW, SW, CPW, 1, 1>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, tilesX, tilesY, chunksPerGrp, G);
  return hipGetLastError();
}

template <int CPW>
hipError_t launch_fc(const FcParams& p, int lutMode, hipStream_t st) {
  const int G = qcnn_stage_group(p.K);
  const int stages = (p.M + G - 1) / G;
  const int stagesPerSplit = (stages + p.msplit - 1) / p.msplit;
  const dim3 grid((p.Ct + NGW * CPW - 1) / (NGW * CPW), p.panels, p.msplit);
  const size_t shm = (size_t)2 * STAGE_BYTES;
  const bool two = min(p.D, p.Cs) > 4;
  auto kern = k_fc_aprx<CPW, 0, 1>;
  if (lutMode == 1 && p.K == 128) kern = two ? k_fc_aprx<CPW, 8, 2> : k_fc_aprx<CPW, 8, 1>;
  if (lutMode == 1 && p.K == 64) kern = two ? k_fc_aprx<CPW, 4, 2> : k_fc_aprx<CPW, 4, 1>;
  if (lutMode == 1 && p.K == 32) kern = two ? k_fc_aprx<CPW, 2, 2> : k_fc_aprx<CPW, 2, 1>;
  if (lutMode == 1 && p.K == 16) kern = two ? k_fc_aprx<CPW, 1, 2> : k_fc_aprx<CPW, 1, 1>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, G, stagesPerSplit);
  return hipGetLastError();
}

}  // namespace

// Tile selection (12 gather waves): the workgroup covers all channels of a group whenever <= 32 float2
// accumulators per wave allow it (every further channel chunk would rebuild the same LUT stages), and
// as many positions as the accumulators then leave room for.  The MFMA builder is instantiated for K in
// {16, 32, 64, 128}; any other K <= 128 runs the exact builder.
hipError_t qk_conv_aprx(const ConvParams& pIn, int lutMode, hipStream_t st) {
  ConvParams p = pIn;
  p.lutF16 = (lutMode == 2) ? 1 : 0;
  if (lutMode == 2) lutMode = 1;
  const int Ctg = p.Ct / p.grp;
  if (Ctg % 4 || p.Cs > QCNN_MAX_CS || p.K > QCNN_MAX_K) return hipErrorInvalidValue;
  if (Ctg % 384 == 0) return launch_conv<1, 1, 1, 32>(p, lutMode, st);    // 12 x 32
  if (Ctg > 192 && Ctg <= 288) return launch_conv<1, 1, 1, 24>(p, lutMode, st);   // 12 x 24
  if (Ctg % 192 == 0) return launch_conv<1, 2, 1, 32>(p, lutMode, st);    // 2 positions x 6 x 32
  if (Ctg % 128 == 0) return launch_conv<1, 3, 1, 32>(p, lutMode, st);    // 3 positions x 4 x 32
  if (Ctg % 96 == 0) return launch_conv<2, 2, 2, 16>(p, lutMode, st);     // 2 strips of 2 x 6 x 16
  if (Ctg % 64 == 0) return launch_conv<1, 6, 2, 16>(p, lutMode, st);     // 3 strips of 2 x 4 x 16
  if (Ctg <= 48) return launch_conv<1, 4, 4, 4>(p, lutMode, st);          // 1 strip of 4 x 12 x 4
  return launch_conv<1, 3, 1, 32>(p, lutMode, st);
}

static int fc_channels_per_wave(int Ct) { return Ct >= 384 ? 32 : (Ct >= 96 ? 8 : 4); }
int qk_fc_channels_per_block(int Ct) { return NGW * fc_channels_per_wave(Ct); }

// p.msplit is chosen by the caller (engine): 1 keeps the reference's summation order.
hipError_t qk_fc_aprx(const FcParams& pIn, int lutMode, hipStream_t st) {
  FcParams p = pIn;
  p.lutF16 = (lutMode == 2) ? 1 : 0;
  if (lutMode == 2) lutMode = 1;
  if (p.Ct % 4 || p.Cs > QCNN_MAX_CS || p.K > QCNN_MAX_K || p.msplit < 1) return hipErrorInvalidValue;
  switch (fc_channels_per_wave(p.Ct)) {
    case 32: return launch_fc<32>(p, lutMode, st);
    case 8: return launch_fc<8>(p, lutMode, st);
    default: return launch_fc<4>(p, lutMode, st);
  }
}

hipError_t qk_sum_partials(const float* partial, float* dst, int msplit, size_t n, int relu, hipStream_t st) {
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_sum_partials, dim3(blocks ? blocks : 1), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(dst), msplit, n4, relu);
  return hipGetLastError();
}

hipError_t qk_permute_rows(const float* src, float* dst, const int* map, int D, int panels, hipStream_t st) {
  const size_t rows = (size_t)panels * D;
  const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_permute_rows, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, map, D, panels);
  return hipGetLastError();
}

hipError_t qk_relu(const float* src, float* dst, size_t n, hipStream_t st) {
  const size_t n4 = n / 4;   // panel rows are 128 floats: always a multiple of 4
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_relu, dim3(blocks ? blocks : 1), dim3(256), 0, st, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n4);
  return hipGetLastError();
}

hipError_t qk_lrn(const float* src, float* dst, int panels, int HW, int C, int lrnSiz, float alp, float bet, float ini,
                  hipStream_t st) {
  const size_t rows = (size_t)panels * HW * C;
  const float coeff = alp / lrnSiz;   // float / int, as src/CaffeEva.cc:1055
  if (lrnSiz == 5 || lrnSiz == 3) {   // streaming kernel: 8 pixels per block
    const size_t pixels = (size_t)panels * HW;
    const dim3 grid((unsigned)((pixels + 7) / 8));
    if (lrnSiz == 5)
      hipLaunchKernelGGL(k_lrn_stream<5>, grid, dim3(256), 0, st, reinterpret_cast<const float4*>(src),
                         reinterpret_cast<float4*>(dst), pixels, C, coeff, -bet, ini);
    else
      hipLaunchKernelGGL(k_lrn_stream<3>, grid, dim3(256), 0, st, reinterpret_cast<const float4*>(src),
                         reinterpret_cast<float4*>(dst), pixels, C, coeff, -bet, ini);
    return hipGetLastError();
  }
  const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_lrn, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, rows, C, lrnSiz, coeff, -bet, ini);
  return hipGetLastError();
}

hipError_t qk_pool(const float* src, float* dst, int panels, int H, int W, int C, int Ho, int Wo, int knl, int stride,
                   int pad, hipStream_t st) {
  const size_t rows = (size_t)panels * Ho * Wo * C;
  if ((rows + 7) / 8 < (size_t)1 << 31) {
    hipLaunchKernelGGL(k_pool4, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), panels, H, W, C, Ho, Wo, knl, stride, pad);
    return hipGetLastError();
  }
  const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_pool, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, panels, H, W, C, Ho, Wo, knl,
                     stride, pad);
  return hipGetLastError();
}

hipError_t qk_softmax(const float* src, float* dst, int panels, int C, hipStream_t st) {
  const size_t shm = ((size_t)C * 32 + 32) * sizeof(float);
  if (shm <= 160 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_softmax_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_softmax_lds, dim3(PANEL / 32, panels), dim3(256), shm, st, src, dst, C);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_softmax, dim3((panels * PANEL + 63) / 64), dim3(64), 0, st, src, dst, panels, C);
  return hipGetLastError();
}

hipError_t qk_top5(const float* prob, uint16_t* out, int n, int C, hipStream_t st) {
  const size_t shm = ((size_t)C * 32 + 2 * 8 * 32) * sizeof(float);
  if (shm <= 160 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_top5_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_top5_lds, dim3(PANEL / 32, panels_of(n)), dim3(256), shm, st, prob, out, n, C);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_top5, dim3((n + 63) / 64), dim3(64), 0, st, prob, out, n, C);
  return hipGetLastError();
}

hipError_t qk_pack_nchw(const float* in, float* dst, int n, int C, int H, int W, hipStream_t st) {
  const int E = C * H * W;
  hipLaunchKernelGGL(k_pack, dim3((E + 63) / 64, panels_of(n)), dim3(256), 0, st, in, dst, n, E, C, H * W, 1);
  return hipGetLastError();
}

hipError_t qk_pack_u8(const uint8_t* in, const float* mean, float* dst, int n, int C, int H, int W, int Hs, int Ws,
                      hipStream_t st) {
  const int E = C * H * W;
  hipLaunchKernelGGL(k_pack_u8, dim3((E + 63) / 64, panels_of(n)), dim3(256), 0, st, in, mean, dst, n, C, H, W, Hs, Ws);
  return hipGetLastError();
}

hipError_t qk_pack_rows(const float* in, float* dst, int n, int E, hipStream_t st) {
  hipLaunchKernelGGL(k_pack, dim3((E + 63) / 64, panels_of(n)), dim3(256), 0, st, in, dst, n, E, 1, E, 0);
  return hipGetLastError();
}

hipError_t qk_unpack_rows(const float* src, float* out, int n, int E, hipStream_t st) {
  hipLaunchKernelGGL(k_unpack, dim3((E + 63) / 64, panels_of(n)), dim3(256), 0, st, src, out, n, E);
  return hipGetLastError();
}
