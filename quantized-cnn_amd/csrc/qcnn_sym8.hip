// qcnn_sym8.hip — k_conv_sym8: the conv table kernel (GetInPdMat src/CaffeEva.cc:1261-1296 fused with
// CalcFeatMap_ConvAprx :760-868) as an EIGHT-wave symmetric workgroup with 256 registers per wave.
//
// Why.  A workgroup's accumulators — 128 images x (position, channel) pairs in registers — bound its output tile, and the
// tile's receptive field is what it must build tables for: with 16 waves x 128 registers (k_conv_aprx: 12 gather waves x
// 64 accumulator registers = 384 pairs; k_conv_sym: 16 x 64 = 512) a 384-channel 3x3 layer gets ONE position per
// workgroup and builds every source pixel's tables 9 times (AlexNet conv3: 8.1 after clipping), a 192-channel one two
// positions (conv4: 5.5).  The register file of a CU is the same 512 KB however it is cut: EIGHT waves of 256 registers,
// every wave building AND gathering, spend 192 of them on accumulators — 8 x 96 = 768 pairs, twice k_conv_aprx's — because
// the per-wave overhead (look-up temporaries, operands, offsets) is paid 8 times instead of 16 and no wave idles in a
// builder role.  Tiles: 384 channels x 1x2 (6 builds per position instead of 9), 256 x 1x3 (5 instead of 9), 192 x 2x2
// (4 instead of 6), 128 x 2x3 (7 instead of 9 for a 5x5 kernel).  A stage then serves twice the look-ups, i.e. the fixed
// cost of a stage (64 KB of LDS stores, the matrix instructions, the barrier) is spread over twice the work.
//
// How.  Same stage machine as k_conv_sym (qcnn_kernels.hip): two 64 KB LUT stages in LDS, one s_barrier per stage, stage
// s + 1 multiplied out while stage s is gathered; same table entries in the same (kh, kw, m) order per output, so the
// results are BIT-IDENTICAL to the tile kernels.  Per wave and stage: 8 of the 64 result tiles (two image tiles x four row
// tiles: 16 v_mfma_f32_16x16x4_f32 for 8-dim sub-spaces, 32 ds_write_addtid_b32 behind 8 M0 writes) and 96 (position,
// channel) look-ups in twelve blocks of four ds_read_b128.  What makes 192 accumulator registers fit: the row offsets of a
// block (8 bytes per wave half) are read from the program row in LDS two blocks before the block that needs them (one
// ds_read_b64 into one of two fixed register pairs) instead of a whole stage's offsets sitting in registers (k_conv_aprx:
// two sets); ONE operand set.  With two waves per SIMD nobody hides an LDS round trip per block, so the blocks of a position
// are software-pipelined inside ONE asm statement (qcnn_sym8_gather.h, generated): three sets of two-read temporaries
// (v[232:255]) rotate, the reads of blocks k + 1, k + 2 fly while block k is accumulated, every wait is a count.  The code
// book comes in operand order (ConvParams::ctrd8: one 16-byte load per k-step).  Measured and dropped
// (profiles/r4_sym8/experiments): the two waves of a SIMD in opposite phase order; the build interleaved into the look-ups.
#include "qcnn_kernels.h"
#include "qcnn_dev.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <queue>
#include <type_traits>
#include <utility>
#include <vector>

#ifndef S8_VAR
#define S8_VAR 0      // compile-time, variant builds only (scripts/build_variant.sh -DS8_VAR=n, scripts/variants_sym8.sh; results wrong):
                      // 1 no LUT stores, 2 no matrix instructions, 4 no look-ups.  0 = the production object: every `#if S8_VAR` below drops out
#endif

#ifndef S8_ANTI
#define S8_ANTI 0     // compile-time, variant builds only: 1 = the two waves of a SIMD in OPPOSITE phases (waves 4 .. 7 gather while waves 0 .. 3
                      // build and vice versa: the same loop body, the stage barrier moved between build and look-ups for the second set)
#endif

#ifndef S8_ROWMAJOR
#define S8_ROWMAJOR 0 // compile-time, variant builds only: 1 = the f32 table ROW-MAJOR like the fp16 table (512-byte rows, image quads XOR-swizzled
                      // by the row's index in its tile), product transposed, ONE ds_write_b128 per result tile instead of four add-TID stores
#endif

#ifdef S8_TRACE
// Debug build only (scripts/trace_sym8.py): every wave of workgroup `s8_trace_block` sums, in scalar registers, the cycles
// between its phase marks — period start / build done (matrix instructions + stores issued, next operand loads issued) /
// look-ups done / barrier left — and writes the sums once at the end (marks cost an s_memtime + s_waitcnt lgkmcnt(0) each).
__device__ unsigned long long s8_trace_buf[8 * 8];
__device__ int s8_trace_block = 0;
#define S8_DECL unsigned long long s8_t = __builtin_readcyclecounter(), s8_sum[6] = {0, 0, 0, 0, 0, 0}
#define S8_T(k, s) do { const unsigned long long now_ = __builtin_readcyclecounter(); s8_sum[k] += now_ - s8_t; s8_t = now_; } while (0)
#define S8_DUMP do { if ((int)blockIdx.x == s8_trace_block && blockIdx.y == 0 && (threadIdx.x & 63) == 0) \
    for (int k_ = 0; k_ < 6; ++k_) s8_trace_buf[(threadIdx.x >> 6) * 8 + k_] = s8_sum[k_]; } while (0)
extern "C" int qcnn_debug_trace8_read(unsigned long long* host, int block) {
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(s8_trace_buf), sizeof(unsigned long long) * 8 * 8);
  if (e != hipSuccess) return 1;
  e = hipMemcpyToSymbol(HIP_SYMBOL(s8_trace_block), &block, sizeof(int));
  return e == hipSuccess ? 0 : 1;
}
#else
#define S8_DECL do {} while (0)
#define S8_T(k, s) do {} while (0)
#define S8_DUMP do {} while (0)
#endif

namespace {

constexpr int NW8 = 8;                              // waves per workgroup (2 per SIMD: 256 registers each)
constexpr uint32_t PROG8_LDS = 2u * STAGE_BYTES;    // three program-row buffers behind the two LUT stages
constexpr uint32_t PROG8_BUF = 2048u;

#include "qcnn_sym8_gather.h"

// the lane index from the execution mask (two VALU instructions): where a value derived from it is needed once per stage
// period or less, re-deriving beats keeping it — the 96-pair instantiations have no register to spare
// (volatile asm: the builtin form is loop-invariant to the compiler, which hoists it and keeps what is derived from it)
__device__ __forceinline__ int lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// wave 0 test on an opaque scalar copy (the compiler otherwise carries the uniform condition as a lane mask in a VGPR across
// the stage loop), and the LDS address of a wave half's block of a program row from the lane index of the moment
__device__ __forceinline__ bool is_wave0(int wave) {
  int w = wave;
  asm volatile("" : "+s"(w));
  return w == 0;
}
__device__ __forceinline__ uint32_t my_blk(int wave, int blkBytes) {
  return PROG8_LDS + (uint32_t)(wave * 2 + (lane_now() >> 5)) * (uint32_t)blkBytes;
}

// operands of one stage for this wave: code-book tiles of its four row tiles, activation tiles of its two image tiles
template <int KS>
struct Ops8 {
  float a[4][KS];
  float b[2][KS];
};
template <int KS>
__device__ __forceinline__ void ops8_load(Ops8<KS>& o, const char* __restrict__ xbase, uint32_t xoff0, uint32_t bLane,
                                          const float* __restrict__ ctrd8, int Cs, int m, uint32_t laneA8, int rt0) {
  // code book in operand order (ConvParams::ctrd8): the four row tiles of a k-step are ONE 16-byte load per lane (issuing a
  // vector memory instruction costs ~30 cycles of the wave's time: 2 + 4 loads per stage instead of 8 + 4)
  const char* __restrict__ cbU = reinterpret_cast<const char*>(ctrd8) + ((size_t)m * 2 + (rt0 >> 2)) * KS * 1024;   // uniform
  const char* __restrict__ xbU = xbase + xoff0 + (uint32_t)(m * Cs) * XROWB;       // uniform
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(cbU + ks * 1024 + laneA8);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.a[i][ks] = a4[i];
#pragma unroll
    for (int t = 0; t < 2; ++t) o.b[t][ks] = *reinterpret_cast<const float*>(xbU + (uint32_t)(ks * 4) * XROWB + bLane + t * 64);
  }
}
template <int KS>
__device__ __forceinline__ f32x4 ops8_tile(const Ops8<KS>& o, int t, int i) {
#if S8_VAR & 2
  return f32x4{o.a[i][0], o.b[t][0], o.a[i][KS - 1], o.b[t][KS - 1]};
#endif
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 c = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][0], o.b[t][0], zero, 0, 0, 0);
  if (KS > 1) c = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][KS - 1], o.b[t][KS - 1], c, 0, 0, 0);
  return c;
}
// the wave's eight tiles -> stage buffer; mA / mB = LDS byte address of (buffer, image tile, first row tile) of its two
// image tiles.  The four stores of a tile go out behind ONE M0 write, in the shadow of the next tile's matrix instructions.
#if S8_VAR & 1
#define S8_ST(I, v, m) asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(m))
#else
#define S8_ST(I, v, m) store_tile_all<I>(v, m)
#endif
template <int KS>
__device__ __forceinline__ void ops8_store(const Ops8<KS>& o, uint32_t mA, uint32_t mB) {
  const f32x4 v0 = ops8_tile<KS>(o, 0, 0);
  const f32x4 v1 = ops8_tile<KS>(o, 0, 1);
  __builtin_amdgcn_sched_barrier(0);

  const f32x4 v2 = ops8_tile<KS>(o, 0, 2);
  S8_ST(0, v0, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v3 = ops8_tile<KS>(o, 0, 3);
  S8_ST(1, v1, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v4 = ops8_tile<KS>(o, 1, 0);
  S8_ST(2, v2, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v5 = ops8_tile<KS>(o, 1, 1);
  S8_ST(3, v3, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v6 = ops8_tile<KS>(o, 1, 2);
  S8_ST(0, v4, mB);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v7 = ops8_tile<KS>(o, 1, 3);
  S8_ST(1, v5, mB);
  __builtin_amdgcn_sched_barrier(0);
  S8_ST(2, v6, mB);
  S8_ST(3, v7, mB);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- fp16 table storage (QCNN_OPT_LUT_MODE = 2, BASELINE configs[4]).  A stage is 128 rows x 128 images x 2 B = 32 KB: row
// slot s (the same slot numbering as the f32 table: qcnn_row_slot) starts at byte 256 s of the stage buffer, and the eight
// bytes of image quad q (images 4q .. 4q + 3) sit at position q ^ (s & 15) of the row.  The product is taken TRANSPOSED
// (A = the activations, B = the code words: the two operands of a 16x16x4 instruction have the same lane layout, so this
// is the order of the arguments only — same products, same k order, the same f32 entries): a lane then holds four
// CONSECUTIVE IMAGES of one code word, rounds them to fp16 (two v_cvt_pk_f16_f32, round to nearest even — the oracle's
// qo_study_mode(1, 0) rounding) and stores them with ONE ds_write_b64 — the eight bytes a look-up lane reads with one
// ds_read_b64.  Banks: the 16 lanes of a store group hold 16 different code words of a row tile, whose slots differ in
// their low four bits, so the XOR spreads them over all 32 banks; the 32 lanes of a read group walk one row.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t ROW16B = 256u;                   // bytes of a table row of 128 fp16 entries
// byte offset inside a stage buffer of (row tile 0 of the wave, image tile `it`, this lane's code word and image quad)
__device__ __forceinline__ uint32_t st16_addr(int rt0, int it, uint32_t li, uint32_t lk, uint32_t sw) {
  const uint32_t rr = li ^ (sw << 2);                                   // the row of a tile this lane's code-book operand holds
  const uint32_t slotLow = ((rr & 3u) << 2) | (rr >> 2);                // qcnn_row_slot inside the tile
  return ((uint32_t)rt0 * 16u + slotLow) * ROW16B + ((((uint32_t)it * 4u + lk) ^ slotLow) << 3);
}
template <int I>
__device__ __forceinline__ void st16(char* lds, const f32x4& v, uint32_t adr) {
  const f16x2 lo = __builtin_convertvector(f32x2{v[0], v[1]}, f16x2), hi = __builtin_convertvector(f32x2{v[2], v[3]}, f16x2);
  *reinterpret_cast<uint2*>(lds + adr + I * 16 * (int)ROW16B) = uint2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
}
// row-major f32 table (S8_ROWMAJOR): row slot s at byte 512 s, the 16 bytes of image quad q at position q ^ key(s), key = the row's
// index inside its tile (its low three bits differ over the eight lanes of a ds_write_b128 group: conflict-free)
__device__ __forceinline__ uint32_t st32_addr(int rt0, int it, uint32_t li, uint32_t lk, uint32_t sw) {
  const uint32_t rr = li ^ (sw << 2);
  const uint32_t slotLow = ((rr & 3u) << 2) | (rr >> 2);
  return ((uint32_t)rt0 * 16u + slotLow) * 512u + ((((uint32_t)it * 4u + lk) ^ rr) << 4);
}
template <int I>
__device__ __forceinline__ void st32(char* lds, const f32x4& v, uint32_t adr) {
  *reinterpret_cast<f32x4*>(lds + adr + I * 16 * 512) = v;
}
template <int KS>
__device__ __forceinline__ f32x4 ops8_tile_t(const Ops8<KS>& o, int t, int i) {
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 c = __builtin_amdgcn_mfma_f32_16x16x4f32(o.b[t][0], o.a[i][0], zero, 0, 0, 0);
  if (KS > 1) c = __builtin_amdgcn_mfma_f32_16x16x4f32(o.b[t][KS - 1], o.a[i][KS - 1], c, 0, 0, 0);
  return c;
}
// the wave's eight tiles -> fp16 stage buffer; aA / aB = st16_addr of its two image tiles (+ the buffer's base)
template <int KS>
__device__ __forceinline__ void ops8_store16(char* lds, const Ops8<KS>& o, uint32_t aA, uint32_t aB) {
  const f32x4 v0 = ops8_tile_t<KS>(o, 0, 0);
  const f32x4 v1 = ops8_tile_t<KS>(o, 0, 1);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v2 = ops8_tile_t<KS>(o, 0, 2);
  st16<0>(lds, v0, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v3 = ops8_tile_t<KS>(o, 0, 3);
  st16<1>(lds, v1, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v4 = ops8_tile_t<KS>(o, 1, 0);
  st16<2>(lds, v2, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v5 = ops8_tile_t<KS>(o, 1, 1);
  st16<3>(lds, v3, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v6 = ops8_tile_t<KS>(o, 1, 2);
  st16<0>(lds, v4, aB);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v7 = ops8_tile_t<KS>(o, 1, 3);
  st16<1>(lds, v5, aB);
  __builtin_amdgcn_sched_barrier(0);
  st16<2>(lds, v6, aB);
  st16<3>(lds, v7, aB);
  __builtin_amdgcn_sched_barrier(0);
}

template <int KS>
__device__ __forceinline__ void ops8_store32(char* lds, const Ops8<KS>& o, uint32_t aA, uint32_t aB) {
  const f32x4 v0 = ops8_tile_t<KS>(o, 0, 0);
  const f32x4 v1 = ops8_tile_t<KS>(o, 0, 1);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v2 = ops8_tile_t<KS>(o, 0, 2);
  st32<0>(lds, v0, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v3 = ops8_tile_t<KS>(o, 0, 3);
  st32<1>(lds, v1, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v4 = ops8_tile_t<KS>(o, 1, 0);
  st32<2>(lds, v2, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v5 = ops8_tile_t<KS>(o, 1, 1);
  st32<3>(lds, v3, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v6 = ops8_tile_t<KS>(o, 1, 2);
  st32<0>(lds, v4, aB);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v7 = ops8_tile_t<KS>(o, 1, 3);
  st32<1>(lds, v5, aB);
  __builtin_amdgcn_sched_barrier(0);
  st32<2>(lds, v6, aB);
  st32<3>(lds, v7, aB);
  __builtin_amdgcn_sched_barrier(0);
}

// the 96 look-ups of this wave in stage `c`: NP positions x CPW channels, one pipelined asm statement per position (a
// position that does not look at the stage's pixel is skipped inside it); blk = LDS byte address of the wave half's block
// of the stage's program row ([position][CPW / 2] uint16)
// MODE: 0 = f32 table, f32 sums (gpos*); 1 = fp16 table, f32 sums (hpos*); 2 = fp16 table, packed fp16 sums (apos*: AccT = uint32_t)
template <int CPW, int P, int MODE, typename AccT>
__device__ __forceinline__ void gather8_pos(AccT* acc, uint32_t blk, uint32_t stage, int ok) {
  constexpr int B = CPW / 8;
  static_assert(B == 2 || B == 3 || B == 4 || B == 6, "blocks per position");
  if constexpr (MODE == 2) {                             // fp16 table, fp16 sums: ds_read_b64 + two v_pk_add_f16
    if constexpr (B == 6) apos6<P * CPW>(acc, blk, stage, ok);
    else if constexpr (B == 4) apos4<P * CPW>(acc, blk, stage, ok);
    else if constexpr (B == 3) apos3<P * CPW>(acc, blk, stage, ok);
    else apos2<P * CPW>(acc, blk, stage, ok);
  } else if constexpr (MODE == 1) {                      // fp16 table: ds_read_b64 + v_fma_mix_f32 (hpos*, same generator)
    if constexpr (B == 6) hpos6<P * CPW>(acc, blk, stage, ok);
    else if constexpr (B == 4) hpos4<P * CPW>(acc, blk, stage, ok);
    else if constexpr (B == 3) hpos3<P * CPW>(acc, blk, stage, ok);
    else hpos2<P * CPW>(acc, blk, stage, ok);
  } else {
    if constexpr (B == 6) gpos6<P * CPW>(acc, blk, stage, ok);
    else if constexpr (B == 4) gpos4<P * CPW>(acc, blk, stage, ok);
    else if constexpr (B == 3) gpos3<P * CPW>(acc, blk, stage, ok);
    else gpos2<P * CPW>(acc, blk, stage, ok);
  }
}
template <int TH, int TW, int CPW, int MODE, typename AccT, int... Ps>
__device__ __forceinline__ void gather8_all(AccT (&acc)[TH * TW][CPW], uint32_t blk, uint32_t stage, const int (&ok)[TH * TW],
                                            std::integer_sequence<int, Ps...>) {
  (gather8_pos<CPW, Ps, MODE, AccT>(&acc[Ps][0], blk, stage, ok[Ps]), ...);
}
template <int TH, int TW, int CPW, int MODE = 0, typename AccT = f32x2>
__device__ __forceinline__ void gather8(AccT (&acc)[TH * TW][CPW], uint32_t blk, const StagePos& c, int knl,
                                        const int (&rowStart)[TH], const int (&colStart)[TW], uint32_t stage, int live) {
  constexpr int NP = TH * TW;
  int ok[NP];
  int colOk[TW];
#pragma unroll
  for (int dx = 0; dx < TW; ++dx) colOk[dx] = in_range(c.wi - colStart[dx], knl);
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) {
    const int rowOk = live & in_range(c.hi - rowStart[dy], knl);
#pragma unroll
    for (int dx = 0; dx < TW; ++dx) ok[dy * TW + dx] = uni(rowOk & colOk[dx]);
  }
#if !(S8_VAR & 4)
  gather8_all<TH, TW, CPW, MODE, AccT>(acc, blk, stage, ok, std::make_integer_sequence<int, NP>{});
#endif
}

// packed fp16 sums: two copies of fp16(b) / the two halves back as floats
__device__ __forceinline__ uint32_t pk16(float b) {
  const f16x2 h = {(_Float16)b, (_Float16)b};
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ f32x2 unpk16(uint32_t w) {
  const f16x2 h = __builtin_bit_cast(f16x2, w);
  return f32x2{(float)h[0], (float)h[1]};
}

// SLIDE (the eight-wave form of k_conv_aprx<.., SLIDE>): the workgroup owns a SEGMENT of output rows of a strip of TW output
// columns and sweeps the source rows under it; TH = ceil(knl / stride) accumulator SLOTS per column hold the output rows
// whose windows contain the current source row — when a window closes its sums are stored and the slot restarts from the
// bias TH rows further down.  Every source pixel of the strip is built once per segment: 3 table builds per output position
// for a 3x3 / 1 layer with 192 or 256 channels per workgroup (2x2 / 1x3 tiles: 4 and 5), 2 with 128 channels (two columns),
// 5 for a 5x5 / 1 layer with 128 (2x3 tile: 7).  Positions are [slot][column] — the tile kernel's [row][column] with a
// slot's first source row (xq) in the place of a tile row's; program rows are indexed by the source row modulo TH * stride.
// MODE 1, 2: fp16 table storage (see st16_addr): 32 KB stages of 256-byte rows, ds_read_b64 look-ups; the program rows then hold
// (slot << 8) | ((slot & 15) << 3) instead of slot * 64 (k_build_program8 with f16 = 1).  MODE 1 keeps fp32 sums; MODE 2 keeps the
// running sums as packed fp16 too (the accumulate half of the configs[4] study): half the accumulator registers, so a wave owns
// 192 (position, channel) pairs — twice the tile, a third fewer table builds per output position
template <int CPW, int TH, int TW, int KS, bool SLIDE = false, int MODE = 0>
__global__ __launch_bounds__(NW8 * 64) void k_conv_sym8(ConvParams p, int tilesX, int tilesY, int chunks) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr bool F16 = MODE >= 1, ACC16 = MODE == 2;
  using AccT = std::conditional_t<ACC16, uint32_t, f32x2>;
  constexpr int NP = TH * TW, HC = CPW / 2;
  constexpr int PAIRS = ACC16 ? 192 : 96;
  static_assert(CPW % 8 == 0 && NP * CPW <= PAIRS && (SLIDE || NP * CPW == PAIRS) && !(SLIDE && ACC16),
                "96 (position, channel) pairs of four images per lane (192 with packed fp16 sums) = 192 accumulator registers");
  constexpr int BLKB = NP * CPW;                       // bytes of a wave half's block of a program row ([NP][HC] uint16)
  constexpr int ROWB = NW8 * 2 * BLKB;                 // bytes of the workgroup's program row of one entry (<= 1536; 3072 with fp16 sums)
  constexpr uint32_t PBUF = ROWB > (int)PROG8_BUF ? 3072u : PROG8_BUF;   // bytes of one of the three program-row buffers
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  // blockIdx.x = tile rank (heaviest first) * panels + panel; a launch that cannot fill the chip with whole tiles (one GPU's share
  // of a sharded batch: ConvParams::splitZ > 1, tile form) cuts EVERY tile into splitZ workgroups that take consecutive slices of
  // its stage sequence and write partial sums, which k_conv_sum adds in slice order (as k_conv_aprx does)
  int rank = (int)(blockIdx.x / (unsigned)p.panels);
  const int panel = (int)(blockIdx.x % (unsigned)p.panels);
  int slice = 0, slices = 1;
  if (!SLIDE && p.splitZ > 1) {
    slices = p.splitZ;
    slice = rank % slices;
    rank = rank / slices;
  }
  int ty = 0, tx = 0, segBeg = 0, segEnd = 0;
  if constexpr (SLIDE) {
    // rank = segment-major unit (longest segments first): segment x strip of TW output columns
    const unsigned colGroups = (unsigned)(p.Wo + TW - 1) / TW;
    const int seg = (int)((unsigned)rank / colGroups);
    tx = (int)((unsigned)rank % colGroups);
    segBeg = p.segBeg[seg]; segEnd = p.segBeg[seg + 1];
  } else {
    tile_of_rank(rank, tilesY, tilesX, ty, tx);
  }
  const int grp = (int)blockIdx.y / chunks, chunk = (int)blockIdx.y % chunks;
  const int Cg = p.Cin / p.grp, Ctg = p.Ct / p.grp;
  const int M = p.M;
  const int ho0 = SLIDE ? segBeg : ty * TH, wo0 = tx * TW;
  const int hoL = SLIDE ? segEnd - 1 : min(ho0 + TH, p.Ho) - 1, woL = min(wo0 + TW, p.Wo) - 1;
  const int hiL = max(0, ho0 * p.stride - p.pad), hiU = min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1);
  ConvGeom g;
  g.W = p.W; g.Cin = p.Cin; g.knl = p.knl; g.M = M; g.G = 1; g.rowStride = 0;
  g.pixStride = (uint32_t)p.Cin * (uint32_t)XROWB;
  g.MG = M;
  g.wiL = max(0, wo0 * p.stride - p.pad);
  g.wiU = min(p.W - 1, woL * p.stride - p.pad + p.knl - 1);
  g.slide = SLIDE ? 1 : 0; g.hiL = hiL; g.hiU = hiU; g.period = SLIDE ? TH * p.stride : 1;
  const int cols = g.wiU - g.wiL + 1;
  const int Stot = (hiU - hiL + 1) * cols * g.MG;      // stages of the whole tile; this workgroup runs [sBeg, sBeg + S)
  const int sBeg = (int)((long long)Stot * slice / slices);
  const int S = (int)((long long)Stot * (slice + 1) / slices) - sBeg;
  const int Sp = (S + 1) & ~1;
  const StagePos first = {hiL + (sBeg / g.MG) / cols, g.wiL + (sBeg / g.MG) % cols, sBeg % g.MG,
                          SLIDE ? (int)((unsigned)(hiL - (ho0 * p.stride - p.pad)) % (unsigned)(TH * p.stride)) : 0};
  if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();   // the stage addressing assumes the dynamic segment starts at LDS byte 0

  // ---- builder side of this wave: image tiles 2 (wave >> 1), + 1 (both with slot swizzle wave >> 1), row tiles 4 (wave & 1) ..
  const int it0 = (wave >> 1) * 2, rt0 = (wave & 1) * 4, sw = wave >> 1;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t laneA = (lk * 16 + (li ^ ((uint32_t)sw << 2))) * 16;    // byte offset in a 1 KB operand block: rows pre-swizzled for the tiles' slot order (mfma_load)
  const uint32_t bLane = lk * XROWB + (uint32_t)it0 * 64 + li * 4;
  const char* __restrict__ xbase =
      reinterpret_cast<const char*>(p.src + ((size_t)panel * p.H * p.W * p.Cin + (size_t)grp * Cg) * PANEL);
  constexpr bool RM = S8_ROWMAJOR != 0 && MODE == 0;
  const uint32_t mA0 = F16 ? st16_addr(rt0, it0, li, lk, (uint32_t)sw) : RM ? st32_addr(rt0, it0, li, lk, (uint32_t)sw) : (uint32_t)it0 * TILEB + (uint32_t)rt0 * 1024u;
  const uint32_t mB0 = F16 ? st16_addr(rt0, it0 + 1, li, lk, (uint32_t)sw) : RM ? st32_addr(rt0, it0 + 1, li, lk, (uint32_t)sw) : mA0 + TILEB;
  auto build = [&](const Ops8<KS>& o, uint32_t buf) {       // the wave's eight tiles of a stage -> stage buffer at byte `buf`
    if constexpr (F16) ops8_store16<KS>(lds, o, mA0 + buf, mB0 + buf);
    else if constexpr (RM) ops8_store32<KS>(lds, o, mA0 + buf, mB0 + buf);
    else ops8_store<KS>(o, mA0 + buf, mB0 + buf);
  };
  const int Cs = p.Cs;

  // ---- gather side: channels cw0 .. cw0 + CPW - 1 of the group for every position of the tile
  const int half = lane >> 5, quad = lane & 31;
  const int cw0 = (chunk * NW8 + wave) * CPW;
  const int activeI = in_range(cw0, Ctg);
  const int cl0 = cw0 + half * HC;
  const uint32_t laneLds = F16 ? (uint32_t)quad * 8u : RM ? (uint32_t)quad * 16u : ((uint32_t)(quad >> 2) * TILEB | (uint32_t)(quad >> 3) * 64 | (uint32_t)(quad & 3) * 16);
  AccT acc[NP][CPW];
  {
    const float* __restrict__ bp = p.bias + grp * Ctg + (activeI ? cl0 : 0);
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const float b = (slice == 0) ? bp[j] : 0.0f;       // the bias enters the first slice's partial sum only
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        if constexpr (ACC16) { acc[q][2 * j] = pk16(b); acc[q][2 * j + 1] = pk16(b); }     // the start value rounded to fp16 too
        else { acc[q][2 * j] = f32x2{b, b}; acc[q][2 * j + 1] = f32x2{b, b}; }
      }
    }
  }
  // first source row of every tile row's (SLIDE: every slot's current) window, first source column of every tile column's;
  // positions outside the map / slots past the segment get a start that can never match a tap
  int rowStart[TH], colStart[TW];
  int woq[TH];                                         // SLIDE: the output row a slot holds
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) {
    woq[dy] = ho0 + dy;
    rowStart[dy] = (ho0 + dy <= hoL) ? (ho0 + dy) * p.stride - p.pad : -(1 << 28);
  }
#pragma unroll
  for (int dx = 0; dx < TW; ++dx) colStart[dx] = (wo0 + dx < p.Wo) ? (wo0 + dx) * p.stride - p.pad : -(1 << 28);
  const int rfW = (TW - 1) * p.stride + p.knl;
  const int ry0 = ho0 * p.stride - p.pad, rx0 = wo0 * p.stride - p.pad;
  const uint32_t entryB = (uint32_t)(p.grp * chunks) * ROWB;
  const char* __restrict__ progWg = reinterpret_cast<const char*>(p.progS) + (size_t)(grp * chunks + chunk) * ROWB;
  auto rowOf = [&](const StagePos& q, int idx) {       // stages past the end: any existing row
    const StagePos c = (idx < S) ? q : first;
    const int row = SLIDE ? c.ph : c.hi - ry0;
    return progWg + (size_t)(uint32_t)((row * rfW + (c.wi - rx0)) * M + c.mg) * entryB;
  };
  // SLIDE: after the last stage of a source row the positions whose window ends with this row (or with the strip) are stored
  // and their slot restarts from the bias for the output row TH further down
  auto column_end = [&](const StagePos& c, int live) {
   if constexpr (!ACC16) {
    if (!(live && c.wi == g.wiU && c.mg == g.MG - 1)) return;
    // everything lane-dependent is re-derived HERE from an opaque copy of the lane index: hoisted out of the stage loop these
    // values cost the 96-pair instantiations ten spilled registers, reloaded in every stage period (measured +20 % per stage)
    const int laneC = lane_now();
    const uint32_t halfC = (uint32_t)laneC >> 5, quadC = (uint32_t)laneC & 31u;
    // wave-uniform bases + one 32-bit lane offset each (no per-lane 64-bit pointers)
    float* __restrict__ dstU = p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL + (size_t)(grp * Ctg + cw0) * PANEL;
    const float* __restrict__ biasU = p.bias + grp * Ctg + cw0;
    const uint32_t dstLane = halfC * (uint32_t)(HC * PANEL) + 4u * quadC, biasLane = halfC * (uint32_t)HC;
#pragma unroll
    for (int q = 0; q < TH; ++q) {
      if (rowStart[q] > -(1 << 27) && (c.hi - rowStart[q] == p.knl - 1 || c.hi == hiU)) {
#pragma unroll
        for (int dx = 0; dx < TW; ++dx) {
          const bool colReal = wo0 + dx < p.Wo;
          float* __restrict__ o = dstU + (size_t)(woq[q] * p.Wo + wo0 + dx) * p.Ct * PANEL;   // uniform
#pragma unroll
          for (int j = 0; j < HC; ++j) {
            if (colReal) {
              f32x4 v = {acc[q * TW + dx][2 * j].x, acc[q * TW + dx][2 * j].y, acc[q * TW + dx][2 * j + 1].x, acc[q * TW + dx][2 * j + 1].y};
              if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (0.0f < v[e]) ? v[e] : 0.0f;
              }
              *reinterpret_cast<f32x4*>(o + dstLane + j * PANEL) = v;
            }
            const float b = biasU[biasLane + j];
            acc[q * TW + dx][2 * j] = f32x2{b, b}; acc[q * TW + dx][2 * j + 1] = f32x2{b, b};
          }
        }
        woq[q] += TH;
        rowStart[q] = (woq[q] <= hoL) ? rowStart[q] + TH * p.stride : -(1 << 28);
      }
    }
   }
  };
  auto posOf = [&](const StagePos& q, int idx) { return (idx < S) ? q : first; };
  Ops8<KS> ops;
  StagePos c0 = first;
  StagePos c1 = next_pos(c0, g);
  StagePos c2 = next_pos(c1, g);
  // program rows: three LDS buffers, the row of stage t in buffer t % 3, fetched by LDS-DMA two periods before it is read —
  // wave 0 never waits for it at a barrier: it has landed when the operands loaded after it are consumed a period later
  uint32_t rb0 = 0, rb1 = PBUF, rb2 = 2 * PBUF;               // buffers of stages s, s + 1, s + 2
  // prologue: stage 0 -> buffer 0; operands of stage 1; program rows of stages 0 and 1
  ops8_load<KS>(ops, xbase, pixel_off(c0, g), bLane, p.ctrd8, Cs, c0.mg, laneA, rt0);
  build(ops, 0u);
  {
    const StagePos q = posOf(c1, 1);
    ops8_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd8, Cs, q.mg, laneA, rt0);
  }
  if (wave == 0) { idx_row_to_lds<ROWB>(rowOf(c0, 0), PROG8_LDS + rb0, lane); idx_row_to_lds<ROWB>(rowOf(c1, 1), PROG8_LDS + rb1, lane); }
  barrier_after_lds_dma();
  // S8_ANTI: waves 4 .. 7 (the second wave of every SIMD) run half a period ahead in the look-ups: they gather stage t + 1 where
  // waves 0 .. 3 gather stage t, and wait at the stage barrier BETWEEN build and look-ups instead of behind the look-ups — so
  // that one wave of a SIMD multiplies while the other one gathers.  Same buffers, same hazards: within a barrier period set A
  // builds stage s + 1 and gathers stage s, set B gathers stage s and builds stage s + 1.
  constexpr bool ANTI = S8_ANTI != 0 && !SLIDE && MODE == 0;
  // which waves form set B: S8_ANTI = 1 the second wave of every SIMD (waves 4 .. 7: a workgroup's waves go to the SIMDs in cyclic
  // order, so waves w and w + 4 share one), 2 = the odd waves (two whole SIMDs: 0→2→1→3 puts waves 0, 2 on one pair of SIMDs and
  // 1, 3 on the other), 3 = waves 2, 3, 6, 7 (the other pairing of whole SIMDs)
  const int isB = ANTI ? (S8_ANTI == 1 ? (wave >= 4 ? 1 : 0) : S8_ANTI == 2 ? (wave & 1) : ((wave >> 1) & 1)) : 0;
  if (ANTI && isB) gather8<TH, TW, CPW, MODE, AccT>(acc, my_blk(wave, BLKB) + rb0, c0, p.knl, rowStart, colStart, laneLds, activeI);
  auto pick = [&](const StagePos& a, const StagePos& b) { StagePos r; r.hi = isB ? b.hi : a.hi; r.wi = isB ? b.wi : a.wi; r.mg = isB ? b.mg : a.mg; r.ph = isB ? b.ph : a.ph; return r; };
  {
    S8_DECL;
    StagePos cEnd = first;                              // SLIDE: the stage gathered last (its source row may have ended)
    int liveEnd = 0;
    for (int s = 0; s < Sp; s += 2) {
      // ---- period s: stage s + 1 -> buffer 1, gather stage s out of buffer 0
      build(ops, (uint32_t)STAGE_BYTES);
      S8_T(4, s);
      // program row of stage s + 2, AFTER the build (whose counted vmcnt waits would otherwise wait for this fresh transfer)
      // and BEFORE the operand loads (whose wait, a period later, then covers it)
      if (is_wave0(wave)) idx_row_to_lds<ROWB>(rowOf(c2, s + 2), PROG8_LDS + rb2, lane_now());
      __builtin_amdgcn_sched_barrier(0);
      {
        const StagePos q = posOf(c2, s + 2);
        ops8_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd8, Cs, q.mg, laneA, rt0);
      }
      __builtin_amdgcn_sched_barrier(0);
      S8_T(1, s);
      if constexpr (SLIDE) column_end(cEnd, liveEnd);       // what the previous stage finished (its stores have this period to drain)
      if constexpr (ANTI) {
        if (isB) barrier_after_lds_writes();
        gather8<TH, TW, CPW, MODE, AccT>(acc, my_blk(wave, BLKB) + (isB ? rb1 : rb0), pick(c0, c1), p.knl, rowStart, colStart,
                                         laneLds | (uint32_t)(isB ? STAGE_BYTES : 0), activeI & (isB ? in_range(s + 1, S) : 1));
      } else {
        gather8<TH, TW, CPW, MODE, AccT>(acc, my_blk(wave, BLKB) + rb0, c0, p.knl, rowStart, colStart, laneLds, activeI);
      }
      if constexpr (SLIDE) { cEnd = c0; liveEnd = activeI; }
      S8_T(2, s);
      c0 = c1; c1 = c2; c2 = next_pos(c2, g);
      { const uint32_t t = rb0; rb0 = rb1; rb1 = rb2; rb2 = t; }
      if (!ANTI || !isB) barrier_after_lds_writes();
      S8_T(3, s);
      // ---- period s + 1: stage s + 2 -> buffer 0, gather stage s + 1 out of buffer 1
      build(ops, 0u);
      S8_T(4, s + 1);
      if (is_wave0(wave)) idx_row_to_lds<ROWB>(rowOf(c2, s + 3), PROG8_LDS + rb2, lane_now());
      __builtin_amdgcn_sched_barrier(0);
      {
        const StagePos q = posOf(c2, s + 3);
        ops8_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd8, Cs, q.mg, laneA, rt0);
      }
      __builtin_amdgcn_sched_barrier(0);
      S8_T(1, s + 1);
      if constexpr (SLIDE) column_end(cEnd, liveEnd);
      if constexpr (ANTI) {
        if (isB) barrier_after_lds_writes();
        gather8<TH, TW, CPW, MODE, AccT>(acc, my_blk(wave, BLKB) + (isB ? rb1 : rb0), pick(c0, c1), p.knl, rowStart, colStart,
                                         laneLds | (uint32_t)(isB ? 0 : STAGE_BYTES), activeI & in_range(s + 1 + isB, S));
      } else {
        gather8<TH, TW, CPW, MODE, AccT>(acc, my_blk(wave, BLKB) + rb0, c0, p.knl, rowStart, colStart, laneLds | STAGE_BYTES, activeI & in_range(s + 1, S));
      }
      if constexpr (SLIDE) { cEnd = c0; liveEnd = activeI & in_range(s + 1, S); }
      S8_T(2, s + 1);
      c0 = c1; c1 = c2; c2 = next_pos(c2, g);
      { const uint32_t t = rb0; rb0 = rb1; rb1 = rb2; rb2 = t; }
      if (!ANTI || !isB) barrier_after_lds_writes();
      S8_T(3, s + 1);
    }
    S8_DUMP;
    if constexpr (SLIDE) column_end(cEnd, liveEnd);           // the strip's last source row
    // ---- results (SLIDE: every position was stored when its window closed)
    if (activeI && !SLIDE) {
      // final map, or — a slice of a split tile — this slice's slab of partial sums [tile][slice][panel][position][Ct][128]
      const bool part = slices > 1;
      float* __restrict__ dst = part ? p.partial + ((size_t)(rank * slices + slice) * p.panels + panel) * NP * p.Ct * PANEL
                                     : p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int ho = ho0 + q / TW, wo = wo0 + q % TW;
        if (ho < p.Ho && wo < p.Wo) {
          float* o = dst + ((size_t)(part ? q : ho * p.Wo + wo) * p.Ct + grp * Ctg + cl0) * PANEL + 4 * quad;
#pragma unroll
          for (int j = 0; j < HC; ++j) {
            f32x4 v;
            if constexpr (ACC16) {
              const f32x2 lo = unpk16(acc[q][2 * j]), hi = unpk16(acc[q][2 * j + 1]);
              v = f32x4{lo.x, lo.y, hi.x, hi.y};
            } else {
              v = f32x4{acc[q][2 * j].x, acc[q][2 * j].y, acc[q][2 * j + 1].x, acc[q][2 * j + 1].y};
            }
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

// ------------------------------------------------------------------------------------------------------------------
// k_fc_sym8: the FC table kernel (GetInPdMat :1261-1296 fused with CalcFeatMap_FCntAprx :968-1025) with the same eight
// waves of 256 registers.  An FC stage (four sub-spaces of 32 code words) is built once for 128 images and then serves
// 4 x (channels of the workgroup) look-ups: the kernel is its look-up stream.  k_fc_aprx runs that stream at 4.9 cycles per
// row look-up (offset bytes through the vector memory path, expanded with two VALU instructions per dword, blocks of eight
// reads with a round trip each); here the offsets come as uint16 through LDS-DMA like the conv program rows, a wave owns
// 96 channels (768 per workgroup: 6 instead of 11 channel chunks for 4096 outputs) and runs the software-pipelined position
// statements of k_conv_sym8 (a sub-space of the stage = a "position" accumulating into the same registers).
// K = 32, Cs = 4, M a multiple of 4, D = 4 M (AlexNet / VGG-16 fc6, fc7); everything else stays with k_fc_aprx.
// Stage loop, barriers and add-TID stores as in k_conv_sym8; the sub-space axis is split over blockIdx.z like k_fc_aprx's.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FC8_CPW = 96;                               // channels per wave (192 accumulator registers)
constexpr int FC8_SUBB = NW8 * 2 * (FC8_CPW / 2) * 2;     // program bytes of one sub-space for the workgroup: 16 x 48 uint16 = 1536
constexpr uint32_t FC8_ROWBUF = 4u * FC8_SUBB;            // a stage's four sub-spaces: 6 KB

struct FcOps8 {
  float a[4];        // code-book tiles of the wave's four row tiles (two sub-spaces x two tiles of 16 code words)
  float b[2][2];     // activation tiles [image tile][sub-space of the pair]
};
// ctrdF: the code book in operand order [stage][2 halves of the row tiles][64 lanes][4 row tiles] (qk_ctrdf_index)
__device__ __forceinline__ void fc8_load(FcOps8& o, const char* __restrict__ xbase, const float* __restrict__ ctrdF, int stage,
                                         int subA, uint32_t bLane, uint32_t laneA16, int h) {
  const f32x4 a4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ctrdF) + ((size_t)stage * 2 + h) * 1024 + laneA16);
#pragma unroll
  for (int i = 0; i < 4; ++i) o.a[i] = a4[i];
  const char* __restrict__ xbU = xbase + (uint32_t)(subA * 4) * XROWB;           // uniform: dims of the pair's first sub-space
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) o.b[t][j] = *reinterpret_cast<const float*>(xbU + (uint32_t)(j * 4) * XROWB + bLane + t * 64);
}
__device__ __forceinline__ void fc8_store(const FcOps8& o, uint32_t mA, uint32_t mB) {
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  auto tile = [&](int t, int i) { return __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i], o.b[t][i >> 1], zero, 0, 0, 0); };
  const f32x4 v0 = tile(0, 0), v1 = tile(0, 1), v2 = tile(0, 2);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v3 = tile(0, 3);
  S8_ST(0, v0, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v4 = tile(1, 0);
  S8_ST(1, v1, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v5 = tile(1, 1);
  S8_ST(2, v2, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v6 = tile(1, 2);
  S8_ST(3, v3, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v7 = tile(1, 3);
  S8_ST(0, v4, mB);
  __builtin_amdgcn_sched_barrier(0);
  S8_ST(1, v5, mB);
  S8_ST(2, v6, mB);
  S8_ST(3, v7, mB);
  __builtin_amdgcn_sched_barrier(0);
}

// fp16 table storage (see st16_addr): the same eight tiles, product transposed, one ds_write_b64 per tile
__device__ __forceinline__ void fc8_store16(char* lds, const FcOps8& o, uint32_t aA, uint32_t aB) {
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  auto tile = [&](int t, int i) { return __builtin_amdgcn_mfma_f32_16x16x4f32(o.b[t][i >> 1], o.a[i], zero, 0, 0, 0); };
  const f32x4 v0 = tile(0, 0), v1 = tile(0, 1), v2 = tile(0, 2);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v3 = tile(0, 3);
  st16<0>(lds, v0, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v4 = tile(1, 0);
  st16<1>(lds, v1, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v5 = tile(1, 1);
  st16<2>(lds, v2, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v6 = tile(1, 2);
  st16<3>(lds, v3, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v7 = tile(1, 3);
  st16<0>(lds, v4, aB);
  __builtin_amdgcn_sched_barrier(0);
  st16<1>(lds, v5, aB);
  st16<2>(lds, v6, aB);
  st16<3>(lds, v7, aB);
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void fc8_store32(char* lds, const FcOps8& o, uint32_t aA, uint32_t aB) {
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  auto tile = [&](int t, int i) { return __builtin_amdgcn_mfma_f32_16x16x4f32(o.b[t][i >> 1], o.a[i], zero, 0, 0, 0); };
  const f32x4 v0 = tile(0, 0), v1 = tile(0, 1), v2 = tile(0, 2);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v3 = tile(0, 3);
  st32<0>(lds, v0, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v4 = tile(1, 0);
  st32<1>(lds, v1, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v5 = tile(1, 1);
  st32<2>(lds, v2, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v6 = tile(1, 2);
  st32<3>(lds, v3, aA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v7 = tile(1, 3);
  st32<0>(lds, v4, aB);
  __builtin_amdgcn_sched_barrier(0);
  st32<1>(lds, v5, aB);
  st32<2>(lds, v6, aB);
  st32<3>(lds, v7, aB);
  __builtin_amdgcn_sched_barrier(0);
}

// the look-ups of a stage: four sub-spaces x two halves of the wave's 96 channels, each the 24-read position statement of
// k_conv_sym8 accumulating into the same registers
template <int MODE, typename AccT>
__device__ __forceinline__ void fc8_gather(AccT (&acc)[FC8_CPW], uint32_t blk, uint32_t stageBase, int live) {
  const int ok = uni(live);
  if constexpr (MODE == 2) {
    apos6<0 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    apos6<0 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    apos6<1 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    apos6<1 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    apos6<2 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    apos6<2 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    apos6<3 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    apos6<3 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
  } else if constexpr (MODE == 1) {
    hpos6<0 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    hpos6<0 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    hpos6<1 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    hpos6<1 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    hpos6<2 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    hpos6<2 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    hpos6<3 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    hpos6<3 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
  } else {
    gpos6<0 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    gpos6<0 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    gpos6<1 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    gpos6<1 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    gpos6<2 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    gpos6<2 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
    gpos6<3 * FC8_SUBB>(&acc[0], blk, stageBase, ok);
    gpos6<3 * FC8_SUBB + 48>(&acc[48], blk, stageBase, ok);
  }
}

template <int MODE>
__global__ __launch_bounds__(NW8 * 64) void k_fc_sym8(FcParams p, const uint16_t* __restrict__ prog, const float* __restrict__ ctrdF,
                                                       int chunks, int stagesPerSplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr bool F16 = MODE >= 1, ACC16 = MODE == 2;     // fp16 table storage; packed fp16 running sums (per workgroup: the slices of
  using AccT = std::conditional_t<ACC16, uint32_t, f32x2>;   // a split sub-space axis are still added in fp32 by k_sum_partials)
  constexpr int HC = FC8_CPW / 2;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int chunk = blockIdx.x, panel = blockIdx.y, split = blockIdx.z;
  const int stagesAll = p.M / 4;
  const int sBeg = split * stagesPerSplit;
  const int S = min(stagesAll, sBeg + stagesPerSplit) - sBeg;       // stages of this workgroup (>= 1 by the launcher)
  const int Sp = (S + 1) & ~1;
  if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();

  // ---- build side (see k_conv_sym8): image tiles 2 (wave >> 1), + 1; row tiles 4 (wave & 1) .. + 3 = sub-spaces 2 (wave & 1), + 1 of the stage
  const int it0 = (wave >> 1) * 2, h = wave & 1, sw = wave >> 1;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t laneA16 = (lk * 16 + (li ^ ((uint32_t)sw << 2))) * 16;
  const uint32_t bLane = lk * XROWB + (uint32_t)it0 * 64 + li * 4;
  const char* __restrict__ xbase = reinterpret_cast<const char*>(p.src + (size_t)panel * p.D * PANEL);
  constexpr bool RM = S8_ROWMAJOR != 0 && MODE == 0;
  const uint32_t mA0 = F16 ? st16_addr(h * 4, it0, li, lk, (uint32_t)sw) : RM ? st32_addr(h * 4, it0, li, lk, (uint32_t)sw) : (uint32_t)it0 * TILEB + (uint32_t)(h * 4) * 1024u;
  const uint32_t mB0 = F16 ? st16_addr(h * 4, it0 + 1, li, lk, (uint32_t)sw) : RM ? st32_addr(h * 4, it0 + 1, li, lk, (uint32_t)sw) : mA0 + TILEB;
  auto build = [&](const FcOps8& o, uint32_t buf) {
    if constexpr (F16) fc8_store16(lds, o, mA0 + buf, mB0 + buf);
    else if constexpr (RM) fc8_store32(lds, o, mA0 + buf, mB0 + buf);
    else fc8_store(o, mA0 + buf, mB0 + buf);
  };

  // ---- gather side
  const int half = lane >> 5, quad = lane & 31;
  const int cw0 = (chunk * NW8 + wave) * FC8_CPW;
  const int activeI = in_range(cw0, p.Ct);
  const int cl0 = cw0 + half * HC;
  const uint32_t laneLds = F16 ? (uint32_t)quad * 8u : RM ? (uint32_t)quad * 16u : ((uint32_t)(quad >> 2) * TILEB | (uint32_t)(quad >> 3) * 64 | (uint32_t)(quad & 3) * 16);
  AccT acc[FC8_CPW];
#pragma unroll
  for (int c = 0; c < FC8_CPW; ++c) {
    if constexpr (ACC16) acc[c] = 0u; else acc[c] = f32x2{0.0f, 0.0f};
  }
  if (split == 0) {
    const float* __restrict__ bp = p.bias + (activeI ? cl0 : 0);
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      const float b = (cl0 + j < p.Ct) ? bp[j] : 0.0f;
      if constexpr (ACC16) { acc[2 * j] = pk16(b); acc[2 * j + 1] = pk16(b); }
      else { acc[2 * j] = f32x2{b, b}; acc[2 * j + 1] = f32x2{b, b}; }
    }
  }
  // program: [M][chunks][16 wave halves][48] uint16; a stage = four consecutive sub-spaces
  const char* __restrict__ progWg = reinterpret_cast<const char*>(prog) + (size_t)chunk * FC8_SUBB;
  const uint32_t subStride = (uint32_t)chunks * FC8_SUBB;
  auto stageOf = [&](int idx) { return sBeg + min(idx, S - 1); };                 // stages past the end: the last one again
  const uint32_t myBlk = PROG8_LDS + (uint32_t)(wave * 2 + half) * (HC * 2);
  // waves 0 .. 3 each fetch one sub-space's 1536 bytes of a stage's program rows (two DMA instructions)
  auto dma_rows = [&](int stage, uint32_t rowBuf) {
    if (wave < 4) idx_row_to_lds<FC8_SUBB>(progWg + (size_t)(stage * 4 + wave) * subStride, PROG8_LDS + rowBuf + (uint32_t)wave * FC8_SUBB, lane);
  };

  FcOps8 ops;
  uint32_t rb0 = 0, rb1 = FC8_ROWBUF, rb2 = 2 * FC8_ROWBUF;
  fc8_load(ops, xbase, ctrdF, stageOf(0), stageOf(0) * 4 + 2 * h, bLane, laneA16, h);
  build(ops, 0u);
  fc8_load(ops, xbase, ctrdF, stageOf(1), stageOf(1) * 4 + 2 * h, bLane, laneA16, h);
  dma_rows(stageOf(0), rb0);
  dma_rows(stageOf(1), rb1);
  barrier_after_lds_dma();
  for (int s = 0; s < Sp; s += 2) {
    // ---- period s: stage s + 1 -> buffer 1, gather stage s out of buffer 0
    build(ops, (uint32_t)STAGE_BYTES);
    dma_rows(stageOf(s + 2), rb2);
    __builtin_amdgcn_sched_barrier(0);
    fc8_load(ops, xbase, ctrdF, stageOf(s + 2), stageOf(s + 2) * 4 + 2 * h, bLane, laneA16, h);
    __builtin_amdgcn_sched_barrier(0);
    fc8_gather<MODE, AccT>(acc, myBlk + rb0, laneLds, activeI);
    { const uint32_t t = rb0; rb0 = rb1; rb1 = rb2; rb2 = t; }
    barrier_after_lds_writes();
    // ---- period s + 1: stage s + 2 -> buffer 0, gather stage s + 1 out of buffer 1
    build(ops, 0u);
    dma_rows(stageOf(s + 3), rb2);
    __builtin_amdgcn_sched_barrier(0);
    fc8_load(ops, xbase, ctrdF, stageOf(s + 3), stageOf(s + 3) * 4 + 2 * h, bLane, laneA16, h);
    __builtin_amdgcn_sched_barrier(0);
    fc8_gather<MODE, AccT>(acc, myBlk + rb0, laneLds | STAGE_BYTES, activeI & in_range(s + 1, S));
    { const uint32_t t = rb0; rb0 = rb1; rb1 = rb2; rb2 = t; }
    barrier_after_lds_writes();
  }
  if (activeI) {
    float* base = (p.msplit > 1) ? p.partial + (size_t)split * p.panels * p.Ct * PANEL : p.dst;
    float* o = base + ((size_t)panel * p.Ct + cl0) * PANEL + 4 * quad;
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      if (cl0 + j < p.Ct) {
        f32x4 v;
        if constexpr (ACC16) {
          const f32x2 lo = unpk16(acc[2 * j]), hi = unpk16(acc[2 * j + 1]);
          v = f32x4{lo.x, lo.y, hi.x, hi.y};
        } else {
          v = f32x4{acc[2 * j].x, acc[2 * j].y, acc[2 * j + 1].x, acc[2 * j + 1].y};
        }
        if (p.relu && p.msplit == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (0.0f < v[e]) ? v[e] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(o + j * PANEL) = v;
      }
    }
  }
}

// rows ([M][rowStride] slot bytes, FC QkSlots order) -> [M][chunks][8 waves][2 halves][48] pre-scaled uint16 offsets
// offset of row slot `slot` as the look-up statements consume it: f32 table slot * 64 (inside an image tile), fp16 table the
// row's byte offset with its XOR key in bits 3..6 (st16_addr)
__device__ __forceinline__ uint16_t prog_entry(int slot, int f16) {
#if S8_ROWMAJOR
  if (!f16) return (uint16_t)((slot << 9) | (((((slot & 3) << 2) | ((slot >> 2) & 3))) << 4));   // row-major f32 table: row + XOR key (st32_addr)
#endif
  return f16 ? (uint16_t)((slot << 8) | ((slot & 15) << 3)) : (uint16_t)(slot * 64);
}
__global__ __launch_bounds__(256) void k_build_program_fc8(const uint8_t* __restrict__ rows, uint16_t* __restrict__ prog, QkSlots src,
                                                           int Ct, int chunks, size_t n, int f16) {
  const int hc = FC8_CPW / 2, subU16 = chunks * NW8 * 2 * hc;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e % (size_t)subU16), m = (int)(e / (size_t)subU16);
    const int j = r % hc, wh = r / hc;
    const int half = wh & 1, wave = (wh >> 1) % NW8, chunk = (wh >> 1) / NW8;
    const int ch = (chunk * NW8 + wave) * FC8_CPW + half * hc + j;
    const int at = ch < Ct ? qk_slot_entry(src, 0, ch) : -1;
    prog[e] = at >= 0 ? prog_entry(rows[(size_t)m * src.rowStride + at], f16) : (uint16_t)0;
  }
}

// rows (plain table of row slots, [kh][kw][M][rowStride], `src` order) -> program of the eight-wave layout: entry (ry, rx, m)
// holds per (group, channel chunk), wave and wave half ONE block [position][CPW / 2] of pre-scaled uint16 offsets (0 where
// the position has no tap at that pixel or the channel does not exist).  One thread per uint16.
__global__ __launch_bounds__(256) void k_build_program8(const uint8_t* __restrict__ rows, uint16_t* __restrict__ prog, QkSlots src,
                                                        Qk8Config cf, int Ctg, int groups, int knl, int stride, int M, size_t n, int f16) {
  const int hc = cf.cpw / 2, np = cf.th * cf.tw;
  const int blkU16 = np * hc, rowU16 = groups * cf.chunks * NW8 * 2 * blkU16;
  const int rfW = (cf.tw - 1) * stride + knl;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e % (size_t)rowU16);
    const int row = (int)(e / (size_t)rowU16);
    const int m = row % M, pix = row / M;
    const int ry = pix / rfW, rx = pix % rfW;
    const int wh = r / blkU16, r3 = r % blkU16;
    const int pos = r3 / hc, j = r3 % hc;
    const int half = wh & 1, waveG = wh >> 1;
    const int wave = waveG % NW8, gc = waveG / NW8;
    const int chunk = gc % cf.chunks, g = gc / cf.chunks;
    const int ch = (chunk * NW8 + wave) * cf.cpw + half * hc + j;
    // tile: position (dy, dx) looks at tap (ry - dy * stride, rx - dx * stride); sliding: slot dy at tap row (ry - dy * stride)
    // modulo the period th * stride (ry = source row modulo that period)
    const int period = cf.th * stride;
    const int kh = cf.slide ? ((ry - (pos / cf.tw) * stride) % period + period) % period : ry - (pos / cf.tw) * stride;
    const int kw = rx - (pos % cf.tw) * stride;
    uint16_t v = 0;
    if (ch < Ctg && (unsigned)kh < (unsigned)knl && (unsigned)kw < (unsigned)knl) {
      const int at = qk_slot_entry(src, g, ch);
      if (at >= 0) v = prog_entry(rows[(size_t)((kh * knl + kw) * M + m) * src.rowStride + at], f16);
    }
    prog[e] = v;
  }
}

template <int CPW, int TH, int TW, bool SLIDE = false, int MODE = 0>
hipError_t launch_sym8(const ConvParams& p, const Qk8Config& cf, hipStream_t st) {
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH;
  // SLIDE: grid.x = (segments x strips of TW output columns, longest segments first) x panels
  const bool split = !SLIDE && MODE == 0 && p.splitZ > 1 && p.partial != nullptr;     // every tile in p.splitZ slices (splitFrom = 0)
  ConvParams q = p;
  if (!split) { q.splitZ = 1; q.partial = nullptr; }
  const dim3 grid((unsigned)((SLIDE ? p.nSeg * tilesX : tilesX * tilesY * q.splitZ) * p.panels), (unsigned)(p.grp * cf.chunks), 1);
  const size_t shm = (size_t)2 * STAGE_BYTES + 3 * (size_t)(NW8 * 2 * TH * TW * CPW > (int)PROG8_BUF ? 3072 : PROG8_BUF);
  const bool two = std::min(p.Cin / p.grp, p.Cs) > 4;
  auto kern = two ? k_conv_sym8<CPW, TH, TW, 2, SLIDE, MODE> : k_conv_sym8<CPW, TH, TW, 1, SLIDE, MODE>;
  hipError_t e = allow_big_lds(reinterpret_cast<const void*>(kern), (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW8 * 64), shm, st, q, tilesX, tilesY, cf.chunks);
  e = hipGetLastError();
  if (e != hipSuccess || !split) return e;
  return qk_conv_sum(q.partial, p.dst, 0, q.splitZ, p.panels, tilesX, tilesY, TH, TW, p.Ho, p.Wo, p.Ct, p.relu, st);
}

}  // namespace

hipError_t qk_build_program8(const uint8_t* rows, uint16_t* prog, const QkSlots& src, const Qk8Config& cf, int Ctg, int groups,
                             int knl, int stride, int M, hipStream_t st, int f16) {
  const size_t n = qk_conv_sym8_program_bytes(cf, groups, knl, stride, M) / sizeof(uint16_t);
  if (!n) return hipErrorInvalidValue;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(k_build_program8, dim3(grid), dim3(256), 0, st, rows, prog, src, cf, Ctg, groups, knl, stride, M, n, f16);
  return hipGetLastError();
}

// p.nSeg / p.segBeg from qk_conv_sym8_slide_plan, p.progS = the sliding program (qk_build_program8 with the sliding config)
hipError_t qk_conv_sym8_slide(const ConvParams& p, hipStream_t st) {
  const Qk8Config cf = qk_conv_sym8_slide_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K, p.knl, p.stride);
  if (!cf.cpw || p.progS == nullptr || p.ctrd8 == nullptr || p.srcNchw || p.nSeg < 1 || p.nSeg > QK_MAX_SEGS) return hipErrorInvalidValue;
  if (cf.th == 3 && cf.cpw == 16) return launch_sym8<16, 3, 2, true>(p, cf, st);
  if (cf.th == 3 && cf.cpw == 24) return launch_sym8<24, 3, 1, true>(p, cf, st);
  if (cf.th == 3 && cf.cpw == 32) return launch_sym8<32, 3, 1, true>(p, cf, st);
  if (cf.th == 5 && cf.cpw == 16) return launch_sym8<16, 5, 1, true>(p, cf, st);
  return hipErrorInvalidValue;
}

// mode 1: the fp16-storage form (QCNN_OPT_LUT_MODE = 2), mode 2: fp16 storage + fp16 sums (QCNN_OPT_LUT_MODE = 3, the tiles of
// qk_conv_sym8_config16); p.progS then is the program of that configuration built with f16 = 1
hipError_t qk_conv_sym8(const ConvParams& p, hipStream_t st, int mode) {
  const Qk8Config cf = qk_conv_sym8_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K);
  if (!cf.cpw || p.progS == nullptr || p.ctrd8 == nullptr || p.srcNchw) return hipErrorInvalidValue;
  if (mode == 2) {
    switch (cf.cpw) {
      case 48: return launch_sym8<48, 2, 2, false, 2>(p, cf, st);
      case 32: return launch_sym8<32, 2, 3, false, 2>(p, cf, st);
      case 24: return launch_sym8<24, 2, 4, false, 2>(p, cf, st);
      case 16: return launch_sym8<16, 3, 4, false, 2>(p, cf, st);
      default: return hipErrorInvalidValue;
    }
  }
  if (mode == 1) {
    switch (cf.cpw) {
      case 48: return launch_sym8<48, 1, 2, false, 1>(p, cf, st);
      case 32: return launch_sym8<32, 1, 3, false, 1>(p, cf, st);
      case 24: return launch_sym8<24, 2, 2, false, 1>(p, cf, st);
      case 16: return launch_sym8<16, 2, 3, false, 1>(p, cf, st);
      default: return hipErrorInvalidValue;
    }
  }
  switch (cf.cpw) {
    case 48: return launch_sym8<48, 1, 2>(p, cf, st);
    case 32: return launch_sym8<32, 1, 3>(p, cf, st);
    case 24: return launch_sym8<24, 2, 2>(p, cf, st);
    case 16: return launch_sym8<16, 2, 3>(p, cf, st);
    default: return hipErrorInvalidValue;
  }
}

bool qk_fc_sym8_shape(int D, int Ct, int M, int Cs, int K) {
  return K == 32 && Cs == 4 && M % 4 == 0 && D == 4 * M && Ct >= 2 * FC8_CPW && Ct % 2 == 0;
}
int qk_fc_sym8_chunks(int Ct) { return (Ct + NW8 * FC8_CPW - 1) / (NW8 * FC8_CPW); }
size_t qk_fc_sym8_program_bytes(int Ct, int M) { return (size_t)M * qk_fc_sym8_chunks(Ct) * FC8_SUBB; }

hipError_t qk_build_program_fc8(const uint8_t* rows, uint16_t* prog, const QkSlots& src, int Ct, int M, hipStream_t st, int f16) {
  const size_t n = qk_fc_sym8_program_bytes(Ct, M) / sizeof(uint16_t);
  const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(k_build_program_fc8, dim3(grid), dim3(256), 0, st, rows, prog, src, Ct, qk_fc_sym8_chunks(Ct), n, f16);
  return hipGetLastError();
}

// p.msplit = workgroups along the sub-space axis (partial sums in p.partial when > 1, reduced by qk_sum_partials)
hipError_t qk_fc_sym8(const FcParams& p, const uint16_t* prog, const float* ctrdF, hipStream_t st, int mode) {
  if (!qk_fc_sym8_shape(p.D, p.Ct, p.M, p.Cs, p.K) || prog == nullptr || ctrdF == nullptr || p.msplit < 1) return hipErrorInvalidValue;
  const int stages = p.M / 4;
  const int per = (stages + p.msplit - 1) / p.msplit;
  const int splits = (stages + per - 1) / per;                 // every workgroup along z has at least one stage
  if (splits != p.msplit) return hipErrorInvalidValue;
  const size_t shm = (size_t)2 * STAGE_BYTES + 3 * FC8_ROWBUF;
  auto kern = mode == 2 ? k_fc_sym8<2> : (mode == 1 ? k_fc_sym8<1> : k_fc_sym8<0>);
  hipError_t e = allow_big_lds(reinterpret_cast<const void*>(kern), (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)qk_fc_sym8_chunks(p.Ct), (unsigned)p.panels, (unsigned)p.msplit), dim3(NW8 * 64), shm, st,
                     p, prog, ctrdF, qk_fc_sym8_chunks(p.Ct), per);
  return hipGetLastError();
}
