// qcnn_half8.hip — k_conv_half8: the eight-wave conv table kernel (GetInPdMat src/CaffeEva.cc:1261-1296 fused with
// CalcFeatMap_ConvAprx :760-868) on HALF PANELS of 64 images.
//
// Why.  k_conv_sym8 (qcnn_sym8.hip) holds 768 (position, channel) sums of 128 images per workgroup; its stage — the table of one
// (source pixel, sub-space) for 128 images: 64 result tiles, 64 KB of LDS stores — costs 2540 cycles before the first look-up, and
// the halo of its small tiles makes every pixel's table be built 3.7 - 6.3 times (8 - 9 for 512 channels in two chunks).  The
// register file is the same 512 KB however it is cut: with 64 images per workgroup the same 192 accumulator registers per wave
// hold 1536 (position, channel) sums — TWICE the tile: 128 channels 2x3 -> 3x4, 192: 2x2 -> 2x4, 256: 1x3 -> 2x3, 384: 1x2 -> 2x2,
// 512 channels in ONE chunk of 1x3 — and a stage is half the build (32 result tiles, 32 KB).  Per image the look-ups are the same
// LDS bytes (one ds_read_b128 = FOUR rows x 64 images instead of two rows x 128), the build is (2 x builds-per-pixel ratio) x
// half a stage.
//
// How.  The stage machine, the add-TID stores, the operand-order code book (ConvParams::ctrd8) and the generated look-up
// statements (qcnn_sym8_gather.h) are k_conv_sym8's; what differs:
//   * LDS stage = [4 image tiles][128 row slots][16 images] = 32 KB; inside tile t the position of a slot within its aligned group
//     of four is XOR-ed with t (k_conv_sym8: t >> 1 for eight tiles).  ds_read_b128 is serviced in four groups of 16 lanes —
//     {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) —, so a HARDWARE group, not a run of
//     16 consecutive lanes, reads one table row: its 16 lanes are the four tiles x four image quads of that row, which the XOR puts
//     on the four bank quarters.  Lane -> (group g, index i) is lane_group(); group g of a wave owns channels g * CPW / 4 ..
//   * wave w builds image tile w >> 1 (key w >> 1), row tiles 4 (w & 1) .. + 3: four result tiles, 8 matrix instructions for
//     8-dim sub-spaces, 16 add-TID stores.
//   * the look-up statement of a position serves CPW / 4 reads = CPW channels; layers with few channels per group would get short
//     statements (128 channels / 8 waves = 16 = four reads, whose offset fetch and drain cost as much as the reads).  So the eight
//     waves split into WS sets that share the channels and take every WS-th POSITION of the tile (128 channels: 4 waves x 32
//     channels x 2 sets of 6 of the 3x4 positions, interleaved along a tile row so that a stage's valid positions split evenly;
//     the two waves of a SIMD belong to different sets).
// Same table entries added in the same (kh, kw, m) order per output: bit-identical to every other f32 table kernel.
#include "qcnn_kernels.h"
#include "qcnn_dev.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <queue>
#include <utility>
#include <vector>

#ifndef H8_VAR
#define H8_VAR 0      // compile-time, variant builds only (scripts/build_variant.sh -DH8_VAR=n; results wrong, timing only):
                      // 1 no LUT stores, 2 no matrix instructions, 4 no look-ups, 8 no operand loads / program rows after the prologue
#endif

namespace {

constexpr int NW8 = 8;                              // waves per workgroup (2 per SIMD: 256 registers each)
constexpr uint32_t HSTAGE = 4u * TILEB;             // a stage: four image tiles = 32 KB
constexpr uint32_t PROGH_LDS = 2u * HSTAGE;         // three program-row buffers behind the two LUT stages
constexpr uint32_t PROGH_BUF = 3072u;

#include "qcnn_sym8_gather.h"

__device__ __forceinline__ int lane_now_h() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ bool is_wave0_h(int wave) {
  int w = wave;
  asm volatile("" : "+s"(w));
  return w == 0;
}
// hardware read group of a lane for ds_read_b128 and the lane's index inside it (0 .. 15, in lane order)
constexpr uint32_t GROUP_B = 0xF00F0FF0u;           // lanes (mod 32) of the second group: 4-11, 16-19, 28-31
__device__ __forceinline__ void lane_group(int lane, int& g, int& i) {
  const uint32_t l5 = (uint32_t)lane & 31u;
  const uint32_t isB = (GROUP_B >> l5) & 1u;
  const uint32_t below = (1u << l5) - 1u;
  g = (lane >> 5) * 2 + (int)isB;
  i = __popc((isB ? GROUP_B : ~GROUP_B) & below);
}
// LDS byte address of the lane's group's block of a program row ([wave][4 groups][local position][CPW / 4] uint16)
__device__ __forceinline__ uint32_t my_blk_h(int wave, int blkBytes) {
  int g, i;
  lane_group(lane_now_h(), g, i);
  return PROGH_LDS + (uint32_t)(wave * 4 + g) * (uint32_t)blkBytes;
}

// operands of one stage for this wave: code-book tiles of its four row tiles, the activation tile of its image tile
template <int KS>
struct OpsH {
  float a[4][KS];
  float b[KS];
};
template <int KS>
__device__ __forceinline__ void opsh_load(OpsH<KS>& o, const char* __restrict__ xbase, uint32_t xoff0, uint32_t bLane,
                                          const float* __restrict__ ctrd8, int Cs, int m, uint32_t laneA8, int rt0) {
  const char* __restrict__ cbU = reinterpret_cast<const char*>(ctrd8) + ((size_t)m * 2 + (rt0 >> 2)) * KS * 1024;   // uniform
  const char* __restrict__ xbU = xbase + xoff0 + (uint32_t)(m * Cs) * XROWB;       // uniform
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(cbU + ks * 1024 + laneA8);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.a[i][ks] = a4[i];
    o.b[ks] = *reinterpret_cast<const float*>(xbU + (uint32_t)(ks * 4) * XROWB + bLane);
  }
}
template <int KS>
__device__ __forceinline__ f32x4 opsh_tile(const OpsH<KS>& o, int i) {
#if H8_VAR & 2
  return f32x4{o.a[i][0], o.b[0], o.a[i][KS - 1], o.b[KS - 1]};
#endif
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 c = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][0], o.b[0], zero, 0, 0, 0);
  if (KS > 1) c = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][KS - 1], o.b[KS - 1], c, 0, 0, 0);
  return c;
}
// the wave's four tiles -> stage buffer; mA = LDS byte address of (buffer, image tile, first row tile).  The four stores of a
// tile go out behind ONE M0 write, in the shadow of the next tile's matrix instructions.
#if H8_VAR & 1
#define H8_ST(I, v, m) asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(m))
#else
#define H8_ST(I, v, m) store_tile_all<I>(v, m)
#endif
template <int KS>
__device__ __forceinline__ void opsh_store(const OpsH<KS>& o, uint32_t mA) {
  const f32x4 v0 = opsh_tile<KS>(o, 0);
  const f32x4 v1 = opsh_tile<KS>(o, 1);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v2 = opsh_tile<KS>(o, 2);
  H8_ST(0, v0, mA);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 v3 = opsh_tile<KS>(o, 3);
  H8_ST(1, v1, mA);
  __builtin_amdgcn_sched_barrier(0);
  H8_ST(2, v2, mA);
  H8_ST(3, v3, mA);
  __builtin_amdgcn_sched_barrier(0);
}

// look-ups of local position Q of this wave: CPW / 4 reads (four rows x 64 images each); blk = the lane group's block of the
// stage's program row
template <int CPW, int Q>
__device__ __forceinline__ void gatherh_pos(f32x2* acc, uint32_t blk, uint32_t stage, int ok) {
  constexpr int B = CPW / 16;
  static_assert(B == 2 || B == 3 || B == 4 || B == 6, "reads per position: 8, 12, 16 or 24");
  if constexpr (B == 6) gpos6<Q * (CPW / 2)>(acc, blk, stage, ok);
  else if constexpr (B == 4) gpos4<Q * (CPW / 2)>(acc, blk, stage, ok);
  else if constexpr (B == 3) gpos3<Q * (CPW / 2)>(acc, blk, stage, ok);
  else gpos2<Q * (CPW / 2)>(acc, blk, stage, ok);
}
template <int CPW, int NPW, int... Qs>
__device__ __forceinline__ void gatherh_all(f32x2 (&acc)[NPW][CPW / 2], uint32_t blk, uint32_t stage, const int (&ok)[NPW],
                                            std::integer_sequence<int, Qs...>) {
  (gatherh_pos<CPW, Qs>(&acc[Qs][0], blk, stage, ok[Qs]), ...);
}
// validity of the wave's local positions at stage pixel c (scalar arithmetic only: see in_range), then the statements
template <int CPW, int TH, int TW, int WS>
__device__ __forceinline__ void gatherh(f32x2 (&acc)[TH * TW / WS][CPW / 2], uint32_t blk, const StagePos& c, int knl,
                                        const int (&rowStart)[TH], const int (&colStart)[TW], uint32_t stage, int live, int set) {
  constexpr int NP = TH * TW, NPW = NP / WS;
  int okAll[NP];
  int colOk[TW];
#pragma unroll
  for (int dx = 0; dx < TW; ++dx) colOk[dx] = in_range(c.wi - colStart[dx], knl);
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) {
    const int rowOk = live & in_range(c.hi - rowStart[dy], knl);
#pragma unroll
    for (int dx = 0; dx < TW; ++dx) okAll[dy * TW + dx] = rowOk & colOk[dx];
  }
  int ok[NPW];
#pragma unroll
  for (int q = 0; q < NPW; ++q) {
    if constexpr (WS == 1) ok[q] = uni(okAll[q]);
    else {
      // two sets: local position q of set s = tile position 2 q + ((s + row of q) & 1) — a CHECKERBOARD (TW is even), so that any
      // rectangle of valid positions splits over the sets to within one position (column parity alone: to within TH)
      const int odd = (set + (2 * q) / TW) & 1;
      ok[q] = uni(okAll[2 * q] ^ ((okAll[2 * q] ^ okAll[2 * q + 1]) & -odd));
    }
  }
#if !(H8_VAR & 4)
  gatherh_all<CPW, NPW>(acc, blk, stage, ok, std::make_integer_sequence<int, NPW>{});
#endif
}

// CPW channels per wave, TH x TW output tile, WS wave sets (every set: 8 / WS waves x CPW channels, TH * TW / WS positions),
// KS k-steps of four dims.  grid.x = tile rank (heaviest first) x half panels, grid.y = groups x channel chunks.
// SLIDE (k_conv_sym8's sliding form on half panels): the workgroup owns a SEGMENT of output rows of a strip of TW output columns
// and sweeps the source rows under it; TH = ceil(knl / stride) accumulator SLOTS per column hold the output rows whose windows
// contain the current source row — when a window closes its sums are stored and the slot restarts from the bias TH rows further
// down.  Every source pixel of the strip is built once per segment.  Positions are [slot][column]; program rows are indexed by
// the source row modulo TH * stride.  grid.x = (segment x strip, longest segments first) x half panels.
template <int CPW, int TH, int TW, int WS, int KS, bool SLIDE = false>
__global__ __launch_bounds__(NW8 * 64) void k_conv_half8(ConvParams p, int tilesX, int tilesY, int chunks) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NP = TH * TW, NPW = NP / WS, QC = CPW / 4, WPS = NW8 / WS;
  static_assert((WS == 1 || (WS == 2 && TW % 2 == 0)) && NP % WS == 0 && CPW % 16 == 0 && NPW * CPW <= 192 && (SLIDE || NPW * CPW == 192),
                "192 (position, channel) sums of four images per lane = 192 accumulator registers");
  constexpr int BLKB = NPW * QC * 2;                   // bytes of a lane group's block of a program row ([NPW][QC] uint16)
  constexpr int ROWB = NW8 * 4 * BLKB;                 // bytes of the workgroup's program row of one entry: 3072
  static_assert(ROWB <= (int)PROGH_BUF, "program row buffer");
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const unsigned halves = 2u * (unsigned)p.panels;
  const int rank = (int)(blockIdx.x / halves);
  const int hpIdx = (int)(blockIdx.x % halves);
  const int panel = hpIdx >> 1, hp = hpIdx & 1;
  int ty = 0, tx = 0, segBeg = 0, segEnd = 0;
  if constexpr (SLIDE) {
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
  const int S = (hiU - hiL + 1) * cols * g.MG;         // stages of the tile
  const int Sp = (S + 1) & ~1;
  const StagePos first = {hiL, g.wiL, 0, SLIDE ? (int)((unsigned)(hiL - (ho0 * p.stride - p.pad)) % (unsigned)(TH * p.stride)) : 0};
  if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();   // the stage addressing assumes the dynamic segment starts at LDS byte 0

  // ---- builder side of this wave: image tile wave >> 1 of the half panel (slot swizzle key = the tile), row tiles 4 (wave & 1) ..
  const int it = wave >> 1, rt0 = (wave & 1) * 4;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t laneA = (lk * 16 + (li ^ ((uint32_t)it << 2))) * 16;    // byte offset in a 1 KB operand block: rows pre-swizzled for the tile's slot order
  const uint32_t bLane = lk * XROWB + (uint32_t)(hp * 4 + it) * 64 + li * 4;
  const char* __restrict__ xbase =
      reinterpret_cast<const char*>(p.src + ((size_t)panel * p.H * p.W * p.Cin + (size_t)grp * Cg) * PANEL);
  const uint32_t mA0 = (uint32_t)it * TILEB + (uint32_t)rt0 * 1024u;
  const int Cs = p.Cs;

  // ---- gather side: channels cw0 .. cw0 + CPW - 1 of the group (lane group g4: cl0 .. cl0 + QC - 1) for every WS-th position
  int g4, i16;
  lane_group(lane, g4, i16);
  const int set = wave / WPS, cb = wave % WPS;
  const int cw0 = (chunk * WPS + cb) * CPW;
  const int activeI = in_range(cw0, Ctg);
  const int cl0 = cw0 + g4 * QC;
  const uint32_t laneLds = (uint32_t)(i16 >> 2) * TILEB | (uint32_t)(i16 >> 2) * 64u | (uint32_t)(i16 & 3) * 16u;
  f32x2 acc[NPW][CPW / 2];
  {
    const float* __restrict__ bp = p.bias + grp * Ctg + (activeI ? cl0 : 0);
#pragma unroll
    for (int j = 0; j < QC; ++j) {
      const float b = bp[j];
#pragma unroll
      for (int q = 0; q < NPW; ++q) { acc[q][2 * j] = f32x2{b, b}; acc[q][2 * j + 1] = f32x2{b, b}; }
    }
  }
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
  auto rowOf = [&](const StagePos& q, int idx) __attribute__((always_inline)) {       // stages past the end: any existing row
    const StagePos c = (idx < S) ? q : first;
    const int row = SLIDE ? c.ph : c.hi - ry0;
    return progWg + (size_t)(uint32_t)((row * rfW + (c.wi - rx0)) * M + c.mg) * entryB;
  };
  // SLIDE: after the last stage of a source row the positions whose window ends with this row (or with the strip) are stored
  // and their slot restarts from the bias for the output row TH further down.  Everything lane-dependent is re-derived here from
  // the execution mask (hoisted out of the stage loop such values cost k_conv_sym8 ten spilled registers)
  auto column_end = [&](const StagePos& c, int live) __attribute__((always_inline)) {
    if (!(live && c.wi == g.wiU && c.mg == g.MG - 1)) return;
    int gC, iC;
    lane_group(lane_now_h(), gC, iC);
    float* __restrict__ dstU = p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL + (size_t)(grp * Ctg + cw0) * PANEL + hp * 64;
    const float* __restrict__ biasU = p.bias + grp * Ctg + cw0;
    const uint32_t dstLane = (uint32_t)gC * (uint32_t)(QC * PANEL) + 4u * (uint32_t)iC, biasLane = (uint32_t)gC * (uint32_t)QC;
#pragma unroll
    for (int dy = 0; dy < TH; ++dy) {
      if (rowStart[dy] > -(1 << 27) && (c.hi - rowStart[dy] == p.knl - 1 || c.hi == hiU)) {
#pragma unroll
        for (int k = 0; k < TW / WS; ++k) {
          const int q = dy * (TW / WS) + k;                                   // local position of this wave in slot dy
          const int dx = WS * k + ((set + dy) & (WS - 1));
          const bool colReal = wo0 + dx < p.Wo;
          float* __restrict__ o = dstU + (size_t)(woq[dy] * p.Wo + wo0 + dx) * p.Ct * PANEL;   // uniform
#pragma unroll
          for (int j = 0; j < QC; ++j) {
            if (colReal) {
              f32x4 v = {acc[q][2 * j].x, acc[q][2 * j].y, acc[q][2 * j + 1].x, acc[q][2 * j + 1].y};
              if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (0.0f < v[e]) ? v[e] : 0.0f;
              }
              *reinterpret_cast<f32x4*>(o + dstLane + j * PANEL) = v;
            }
            const float b = biasU[biasLane + j];
            acc[q][2 * j] = f32x2{b, b}; acc[q][2 * j + 1] = f32x2{b, b};
          }
        }
        woq[dy] += TH;
        rowStart[dy] = (woq[dy] <= hoL) ? rowStart[dy] + TH * p.stride : -(1 << 28);
      }
    }
  };
  auto posOf = [&](const StagePos& q, int idx) __attribute__((always_inline)) { return (idx < S) ? q : first; };
  OpsH<KS> ops;
  StagePos c0 = first;
  StagePos c1 = next_pos(c0, g);
  StagePos c2 = next_pos(c1, g);
  // program rows: three LDS buffers, the row of stage t in buffer t % 3, fetched by LDS-DMA two periods before it is read
  uint32_t rb0 = 0, rb1 = PROGH_BUF, rb2 = 2 * PROGH_BUF;
  opsh_load<KS>(ops, xbase, pixel_off(c0, g), bLane, p.ctrd8, Cs, c0.mg, laneA, rt0);
  opsh_store<KS>(ops, mA0);
  {
    const StagePos q = posOf(c1, 1);
    opsh_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd8, Cs, q.mg, laneA, rt0);
  }
  if (wave == 0) { idx_row_to_lds<ROWB>(rowOf(c0, 0), PROGH_LDS + rb0, lane); idx_row_to_lds<ROWB>(rowOf(c1, 1), PROGH_LDS + rb1, lane); }
  barrier_after_lds_dma();
  StagePos cEnd = first;                                // SLIDE: the stage gathered last (its source row may have ended)
  int liveEnd = 0;
  for (int s = 0; s < Sp; s += 2) {
    // ---- period s: stage s + 1 -> buffer 1, gather stage s out of buffer 0
    opsh_store<KS>(ops, mA0 + HSTAGE);
#if !(H8_VAR & 8)
    if (is_wave0_h(wave)) idx_row_to_lds<ROWB>(rowOf(c2, s + 2), PROGH_LDS + rb2, lane_now_h());
    __builtin_amdgcn_sched_barrier(0);
    {
      const StagePos q = posOf(c2, s + 2);
      opsh_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd8, Cs, q.mg, laneA, rt0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SLIDE) column_end(cEnd, liveEnd);       // what the previous stage finished (its stores have this period to drain)
    gatherh<CPW, TH, TW, WS>(acc, my_blk_h(wave, BLKB) + rb0, c0, p.knl, rowStart, colStart, laneLds, activeI, set);
    if constexpr (SLIDE) { cEnd = c0; liveEnd = activeI; }
    c0 = c1; c1 = c2; c2 = next_pos(c2, g);
    { const uint32_t t = rb0; rb0 = rb1; rb1 = rb2; rb2 = t; }
    barrier_after_lds_writes();
    // ---- period s + 1: stage s + 2 -> buffer 0, gather stage s + 1 out of buffer 1
    opsh_store<KS>(ops, mA0);
#if !(H8_VAR & 8)
    if (is_wave0_h(wave)) idx_row_to_lds<ROWB>(rowOf(c2, s + 3), PROGH_LDS + rb2, lane_now_h());
    __builtin_amdgcn_sched_barrier(0);
    {
      const StagePos q = posOf(c2, s + 3);
      opsh_load<KS>(ops, xbase, pixel_off(q, g), bLane, p.ctrd8, Cs, q.mg, laneA, rt0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SLIDE) column_end(cEnd, liveEnd);
    gatherh<CPW, TH, TW, WS>(acc, my_blk_h(wave, BLKB) + rb0, c0, p.knl, rowStart, colStart, laneLds | HSTAGE, activeI & in_range(s + 1, S), set);
    if constexpr (SLIDE) { cEnd = c0; liveEnd = activeI & in_range(s + 1, S); }
    c0 = c1; c1 = c2; c2 = next_pos(c2, g);
    { const uint32_t t = rb0; rb0 = rb1; rb1 = rb2; rb2 = t; }
    barrier_after_lds_writes();
  }
  if constexpr (SLIDE) column_end(cEnd, liveEnd);         // the strip's last source row
  // ---- results: lane (g4, i16) holds channels cl0 .. and images 64 hp + 4 i16 .. + 3 of every local position
  // (SLIDE: every position was stored when its window closed)
  if (activeI && !SLIDE) {
    float* __restrict__ dst = p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL;
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const int pos = WS * q + ((set + (WS * q) / TW) & (WS - 1));     // see gatherh
      const int ho = ho0 + pos / TW, wo = wo0 + pos % TW;
      if (ho < p.Ho && wo < p.Wo) {
        float* o = dst + ((size_t)(ho * p.Wo + wo) * p.Ct + grp * Ctg + cl0) * PANEL + hp * 64 + 4 * i16;
#pragma unroll
        for (int j = 0; j < QC; ++j) {
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
}

// rows (plain table of row slots, [kh][kw][M][rowStride], `src` order) -> program of the half-panel layout: entry (ry, rx, m)
// holds per (group, channel chunk), wave and lane group ONE block [local position][CPW / 4] of pre-scaled uint16 offsets (0
// where the position has no tap at that pixel or the channel does not exist).  One thread per uint16.
__global__ __launch_bounds__(256) void k_build_program_h8(const uint8_t* __restrict__ rows, uint16_t* __restrict__ prog, QkSlots src,
                                                          QkH8Config cf, int Ctg, int groups, int knl, int stride, int M, size_t n) {
  const int qc = cf.cpw / 4, npw = cf.th * cf.tw / cf.ws, wps = NW8 / cf.ws;
  const int blkU16 = npw * qc, rowU16 = groups * cf.chunks * NW8 * 4 * blkU16;
  const int rfW = (cf.tw - 1) * stride + knl;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e % (size_t)rowU16);
    const int row = (int)(e / (size_t)rowU16);
    const int m = row % M, pix = row / M;
    const int ry = pix / rfW, rx = pix % rfW;
    const int wg4 = r / blkU16, r3 = r % blkU16;
    const int q = r3 / qc, j = r3 % qc;
    const int g4 = wg4 & 3, waveG = wg4 >> 2;
    const int wave = waveG % NW8, gc = waveG / NW8;
    const int chunk = gc % cf.chunks, g = gc / cf.chunks;
    const int set = wave / wps, cb = wave % wps;
    const int ch = (chunk * wps + cb) * cf.cpw + g4 * qc + j;
    const int pos = cf.ws * q + ((set + (cf.ws * q) / cf.tw) & (cf.ws - 1));     // wave sets: a checkerboard of the tile (k_conv_half8)
    // tile: position (dy, dx) looks at tap (ry - dy * stride, rx - dx * stride); sliding: slot dy at tap row (ry - dy * stride)
    // modulo the period th * stride (ry = source row modulo that period)
    const int period = cf.th * stride;
    const int kh = cf.slide ? ((ry - (pos / cf.tw) * stride) % period + period) % period : ry - (pos / cf.tw) * stride;
    const int kw = rx - (pos % cf.tw) * stride;
    uint16_t v = 0;
    if (ch < Ctg && (unsigned)kh < (unsigned)knl && (unsigned)kw < (unsigned)knl) {
      const int at = qk_slot_entry(src, g, ch);
      if (at >= 0) v = (uint16_t)(rows[(size_t)((kh * knl + kw) * M + m) * src.rowStride + at] * 64);
    }
    prog[e] = v;
  }
}

template <int CPW, int TH, int TW, int WS, bool SLIDE = false>
hipError_t launch_half8(const ConvParams& p, const QkH8Config& cf, hipStream_t st) {
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH;
  const dim3 grid((unsigned)((SLIDE ? p.nSeg * tilesX : tilesX * tilesY) * 2 * p.panels), (unsigned)(p.grp * cf.chunks), 1);
  const size_t shm = (size_t)2 * HSTAGE + 3 * (size_t)PROGH_BUF;
  const bool two = std::min(p.Cin / p.grp, p.Cs) > 4;
  auto kern = two ? k_conv_half8<CPW, TH, TW, WS, 2, SLIDE> : k_conv_half8<CPW, TH, TW, WS, 1, SLIDE>;
  hipError_t e = allow_big_lds(reinterpret_cast<const void*>(kern), (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW8 * 64), shm, st, p, tilesX, tilesY, cf.chunks);
  return hipGetLastError();
}

}  // namespace

hipError_t qk_build_program_h8(const uint8_t* rows, uint16_t* prog, const QkSlots& src, const QkH8Config& cf, int Ctg, int groups,
                               int knl, int stride, int M, hipStream_t st) {
  const size_t n = qk_conv_half8_program_bytes(cf, groups, knl, stride, M) / sizeof(uint16_t);
  if (!n) return hipErrorInvalidValue;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(k_build_program_h8, dim3(grid), dim3(256), 0, st, rows, prog, src, cf, Ctg, groups, knl, stride, M, n);
  return hipGetLastError();
}

hipError_t qk_conv_half8(const ConvParams& p, hipStream_t st) {
  const QkH8Config cf = qk_conv_half8_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K);
  if (!cf.cpw || p.progS == nullptr || p.ctrd8 == nullptr || p.srcNchw) return hipErrorInvalidValue;
  switch ((p.Ct / p.grp) / cf.chunks) {
    case 128: return launch_half8<32, 3, 4, 2>(p, cf, st);
    case 192: return launch_half8<48, 2, 4, 2>(p, cf, st);
    case 256: return launch_half8<32, 2, 3, 1>(p, cf, st);
    case 384: return launch_half8<48, 2, 2, 1>(p, cf, st);
    case 512: return launch_half8<64, 1, 3, 1>(p, cf, st);
    default: return hipErrorInvalidValue;
  }
}

// p.nSeg / p.segBeg from qk_conv_half8_slide_plan, p.progS = the sliding program (qk_build_program_h8 with the sliding config)
hipError_t qk_conv_half8_slide(const ConvParams& p, hipStream_t st) {
  const QkH8Config cf = qk_conv_half8_slide_config(p.Cin, p.grp, p.Ct, p.M, p.Cs, p.K, p.knl, p.stride);
  if (!cf.cpw || p.progS == nullptr || p.ctrd8 == nullptr || p.srcNchw || p.nSeg < 1 || p.nSeg > QK_MAX_SEGS) return hipErrorInvalidValue;
  switch ((p.Ct / p.grp) / cf.chunks) {
    case 128: return launch_half8<32, 3, 4, 2, true>(p, cf, st);
    case 192: return launch_half8<48, 3, 2, 2, true>(p, cf, st);
    case 256: return launch_half8<32, 3, 2, 1, true>(p, cf, st);
    case 384: return launch_half8<48, 3, 1, 1, true>(p, cf, st);
    case 512: return launch_half8<64, 3, 1, 1, true>(p, cf, st);
    default: return hipErrorInvalidValue;
  }
}
