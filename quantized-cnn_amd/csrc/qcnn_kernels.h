// qcnn_kernels.h — launch wrappers of the gfx950 kernels (internal; the public surface is include/qcnn_hip.h).
//
// Activation layout in HBM ("image-minor panels"): a batch is cut into panels of QCNN_PANEL = 128
// images; feature map l of one panel is a row-major matrix [E_l][128] with E_l = H*W*C elements in the
// reference's NHWC order and the 128 images of the panel innermost: every load/store of a wave is made of
// full 512-byte rows.  The code-word index of a look-up depends on the layer's assignment table only, so
// it is the same for all images: a "gather" is a contiguous LDS row read, not 128 random ones.
//
// LUT stage in LDS: [8 image tiles][128 row slots][16 images] fp32 = 64 KB; inside tile t the position of a
// slot within its aligned group of four is XOR-ed with t >> 1, which puts the four tiles a 16-lane read group
// of ds_read_b128 touches on distinct banks.  A gather lane carries FOUR images and one
// ds_read_b128 serves TWO look-ups: lanes 0-31 read the row of one output channel, lanes 32-63 the row of
// another.  Inside a tile the 16 rows of an MFMA result tile are stored in the order ds_write_addtid_b32
// produces them (qcnn_row_slot).  Assignment tables live on the device as one-byte row slots (LDS offset / 64) in
// the order the gather waves consume them (QkSlots); the conv kernels with K = 128 read pre-scaled uint16 offsets
// through a small per-layer program table built from them (QkProgram).
#ifndef QCNN_KERNELS_H_
#define QCNN_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define QCNN_PANEL 128
#define QCNN_MAX_CS 8          // dims per sub-space supported by the LUT builders
#define QCNN_MAX_K 128         // code words per (pseudo) sub-space a kernel handles (a LUT stage holds 128 rows)
#define QCNN_MAX_K_FILE 256    // code words per sub-space a parameter set may have (uint8 assignments): above 128 the engine cuts the
                               // sub-space into pseudo sub-spaces of <= 127 code words + a zero row (ConvParams::pd)
#define QCNN_ROWS_PAD 256      // bytes of slack after every row-offset table (over-read of the last groups)
#define QCNN_STAGE_ROWS 128    // code-word rows of one LUT stage in LDS
#define QCNN_TILE_BYTES 8192   // one image tile of a stage: 128 rows x 16 images x 4 B
#define QCNN_STAGE_BYTES (8 * QCNN_TILE_BYTES)
#define QCNN_GATHER_WAVES 12   // gather waves of a conv/FC workgroup (of 16; the other four build the stages)

// A LUT stage holds G = qcnn_stage_group(K) consecutive sub-spaces of one source pixel (conv) / of the
// input vector (FC): G * K <= 128 rows;  stage row of a code word = (m % G) * K + assignment  (< 128).
__host__ __device__ static inline int qcnn_stage_group(int K) { return K <= 64 ? QCNN_STAGE_ROWS / K : 1; }
// Slot of stage row r inside an image tile: v_mfma_f32_16x16x4_f32 leaves row 16i + 4q + e of a result tile in
// element e of lane group q, and ds_write_addtid_b32 of element e stores the four lane groups back to back,
// so the row lands in slot 16i + 4e + q (the two 2-bit fields swapped).
__host__ __device__ static inline int qcnn_row_slot(int r) { return (r & 0x70) | ((r & 3) << 2) | ((r >> 2) & 3); }
// pre-scaled LDS byte offset of a stage row inside an image tile (what the assignment tables hold)
__host__ __device__ static inline uint16_t qcnn_row_offset(int r) { return (uint16_t)(qcnn_row_slot(r) * 64); }

// How the 12 gather waves of a workgroup split `C` output channels (conv: the channels of one group; FC: all
// of them), and the device layout of the assignment table that follows from it.  A wave owns `cpw`
// consecutive channels: lanes 0-31 the first cpw/2, lanes 32-63 the second cpw/2.  Per (tap, sub-space) the
// table holds, for every wave slot (group-major, then chunk, then wave) and each half, `hpB` BYTES: one row SLOT
// (qcnn_row_slot of the code word's stage row, < 128) per channel of the half, padded to whole dwords.  Inside a dword
// the four slots sit in the order (e0, e2, e1, e3): (w << 6) & 0x1fc01fc0 then is the pair of pre-scaled 16-bit row
// offsets (e0, e1) the look-up blocks consume, (w >> 2) & 0x1fc01fc0 the pair (e2, e3) — two VALU instructions per
// two look-up reads; the resident table is the size of the reference's uint8 assignment matrices (src/CaffeEva.cc:586,611).
struct QkSlots {
  int cpw;        // channels per gather wave
  int hp;         // slot entries per half-wave and (tap, sub-space): cpw/2 rounded up to even
  int hpB;        // bytes per half-wave and (tap, sub-space): hp rounded up to a multiple of 4
  int chunks;     // workgroups along the channel axis (per group)
  int groups;
  int C;          // channels per group
  int rowStride;  // BYTES per (tap, sub-space) = groups * chunks * 12 * 2 * hpB
};
static inline QkSlots qk_make_slots(int C, int groups, int cpw) {
  QkSlots s;
  s.cpw = cpw; s.C = C; s.groups = groups;
  s.hp = ((cpw / 2 + 1) / 2) * 2;
  s.hpB = (s.hp + 3) / 4 * 4;
  s.chunks = (C + QCNN_GATHER_WAVES * cpw - 1) / (QCNN_GATHER_WAVES * cpw);
  s.rowStride = groups * s.chunks * QCNN_GATHER_WAVES * 2 * s.hpB;
  return s;
}
// byte position of entry e of a half inside the half's hpB bytes (the (e0, e2, e1, e3) order of every dword)
__host__ __device__ static inline int qk_entry_byte(int e) { return (e & ~3) | ((e & 1) << 1) | ((e >> 1) & 1); }
// conv: channels per wave by the channel count of one group (then as many positions per wave as 64-72
// accumulator registers allow: qk_conv_positions)
static inline QkSlots qk_conv_slots(int Ctg, int groups) {
  const int chunks = (Ctg + 383) / 384;
  const int per = (Ctg + chunks - 1) / chunks;       // channels one workgroup has to cover
  const int cpw = per <= 48 ? 4 : per <= 72 ? 6 : per <= 96 ? 8 : per <= 144 ? 12 : per <= 192 ? 16 : per <= 288 ? 24 : 32;
  return qk_make_slots(Ctg, groups, cpw);
}
// output tile (positions per workgroup) that goes with a conv layer's channels per wave: as many positions as 64-72
// accumulator registers allow
__host__ __device__ static inline void qk_conv_tile(int cpw, int* th, int* tw) {
  switch (cpw) {
    case 32: case 24: *th = 1; *tw = 1; break;
    case 16: *th = 1; *tw = 2; break;
    case 12: *th = 1; *tw = 3; break;
    case 8: *th = 2; *tw = 2; break;
    case 6: *th = 2; *tw = 3; break;
    default: *th = 2; *tw = 4; break;
  }
}
// "Program" of a conv layer: the row offsets in the order a workgroup consumes them.  Entry (ry, rx, m) — a source
// pixel RELATIVE to the tile's unclipped receptive field ((TH-1)*stride + knl rows) and a sub-space — holds, for every
// workgroup slice (group, chunk), wave and wave half, ONE contiguous block with the offsets of ALL positions of the
// tile: [pos][hp] uint16 padded to a multiple of 8.  Position (dy, dx) looks at tap (ry - dy*stride, rx - dx*stride);
// taps that do not exist hold 0 (the gather skips them).  One table serves every tile: border tiles just never visit
// the clipped pixels.  A stage then costs ONE contiguous row read for the whole workgroup (it is fetched by one wave
// and handed to the others through LDS) instead of one table row per position and wave.
struct QkProgram {
  int th, tw, np, rfH, rfW;
  int blkU16;      // uint16 per wave half and entry (np * hp rounded up to a multiple of 8)
  int wgRowU16;    // per workgroup slice and entry = 12 * 2 * blkU16
  int rowU16;      // per entry = groups * chunks * wgRowU16
};
__host__ __device__ static inline QkProgram qk_conv_program(const QkSlots& sl, int knl, int stride) {
  QkProgram g;
  qk_conv_tile(sl.cpw, &g.th, &g.tw);
  g.np = g.th * g.tw;
  g.rfH = (g.th - 1) * stride + knl;
  g.rfW = (g.tw - 1) * stride + knl;
  g.blkU16 = (g.np * sl.hp + 7) / 8 * 8;
  g.wgRowU16 = QCNN_GATHER_WAVES * 2 * g.blkU16;
  g.rowU16 = sl.groups * sl.chunks * g.wgRowU16;
  return g;
}
// Sliding variant: slots per workgroup = output rows that look at one source row = ceil(knl / stride) (2 .. 5), and how
// the 12 gather waves split the channels THERE: the slots' accumulators must fit (slots * channels per wave <= 36
// pairs), so a layer may slide with fewer channels per wave — and more workgroups along the channel axis, each building
// the same stages — than its tile kernel uses (AlexNet conv2: 5 slots x 6 channels per wave, two channel chunks).
// ns = 0: the layer cannot slide.
struct QkSlide {
  int ns;         // slots per output column
  int nc;         // output columns per strip: 2 when six slots fit (<= 6 channels per wave: the 3 columns x 1 row a source
                  // row serves become 3 x 2, i.e. 2 instead of 3 table builds per output position for a 3x3 / 1 layer)
  QkSlots sl;
};
static inline QkSlide qk_slide_config(int Ctg, int groups, int knl, int stride) {
  QkSlide r;
  r.ns = 0; r.nc = 1;
  r.sl = qk_make_slots(Ctg, groups, 4);
  const int ns = (knl + stride - 1) / stride;
  if (ns < 2 || ns > 5) return r;
  const int tileCpw = qk_conv_slots(Ctg, groups).cpw;
  const int cands[5] = {16, 12, 8, 6, 4};
  for (int i = 0; i < 5; ++i) {
    const int c = cands[i];
    const bool built = (ns == 2 && c >= 8) || (ns == 3 && c <= 12) || (ns == 4 && c == 8) || (ns == 5 && c <= 6);   // instantiated kernels
    if (c <= tileCpw && ns * c <= 36 && built) {
      r.ns = ns; r.sl = qk_make_slots(Ctg, groups, c);
      r.nc = (ns == 3 && c <= 6) ? 2 : 1;
      return r;
    }
  }
  return r;
}
// program of the sliding variant: entry (source row modulo P = slots * stride, column rx of the strip, m) holds per strip
// column dx and slot q the offsets of tap ((ry - q * stride) mod P, rx - dx * stride) — 0 where that is not a tap
__host__ __device__ static inline QkProgram qk_conv_program_slide(const QkSlots& slS, int ns, int nc, int knl, int stride) {
  QkProgram g;
  g.th = nc; g.tw = ns;                 // positions = [strip column][slot]
  g.np = nc * ns;
  g.rfH = ns * stride;
  g.rfW = (nc - 1) * stride + knl;
  g.blkU16 = (g.np * slS.hp + 7) / 8 * 8;
  g.wgRowU16 = QCNN_GATHER_WAVES * 2 * g.blkU16;
  g.rowU16 = slS.groups * slS.chunks * g.wgRowU16;
  return g;
}
static inline QkSlots qk_fc_slots(int Ct) { return qk_make_slots(Ct, 1, Ct >= 384 ? 32 : (Ct >= 96 ? 8 : 4)); }
// table position (byte index inside one (tap, sub-space) row) of channel c of group g, or -1
__host__ __device__ static inline int qk_slot_entry(const QkSlots& s, int g, int c) {
  if (c < 0 || c >= s.C) return -1;
  const int wave = c / s.cpw, k = c % s.cpw, hc = s.cpw / 2;
  return ((g * s.chunks * QCNN_GATHER_WAVES + wave) * 2 + k / hc) * s.hpB + qk_entry_byte(k % hc);
}
// Workgroups are dispatched in linear order, so the tiles are numbered heaviest first: interior tiles
// (full receptive field = most stages), then the four edges, then the corners.  With a few workgroups per
// CU the last dispatch round is then made of the short border tiles (longest-processing-time-first).
__host__ __device__ static inline void tile_of_rank(int r, int tilesY, int tilesX, int& ty, int& tx) {
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
struct ConvParams {
  int srcNchw;           // 1: src is the network input [nImages][Cin][H][W] read in place by the builders (first layer with
  int nImages;           //    <= 4 channels per group, K = 128 or the exact builder); 0: src is a panel map
  int panel0;            // srcNchw: index of the launch's first panel inside the batch (sub-batches on several streams)
  const float* src;      // [panels][H*W*Cin][128]
  float* dst;            // [panels][Ho*Wo*Ct][128]
  const float* bias;     // [Ct]
  const float* ctrd;     // [M][Cs][K]      (PrepCtrdBuf layout, src/CaffeEva.cc:556-557)
  const float* ctrd8;    // eight-wave symmetric kernel: the code book in its operand order [M][2 halves of the row tiles][k-steps][64 lanes]
                         // [4 row tiles] (qk_ctrd8_index): a lane's four code-book operands of a k-step are ONE 16-byte load; else NULL
  const uint8_t* rows;   // [kh][kw][M][rowStride] (PrepAsmtBuf order, src/CaffeEva.cc:585-586): row slots, QkSlots order
  const uint16_t* prog;  // [rfH][rfW][M][rowU16]: the same offsets in consumption order (QkProgram); panel kernels only
  int H, W, Cin, Ho, Wo, Ct;
  int knl, stride, pad, grp;
  int M, Cs, K;
  int pd;                // sub-spaces per group of input dims: sub-space m covers dims (m / pd) * Cs ...  1 everywhere except layers with
                         // MORE THAN 128 code words per sub-space (the reference's uint8 allows 256): such a sub-space is pd = ceil(K / 127)
                         // pseudo sub-spaces of <= 127 code words + one all-zero row each; an assignment names its code word in one of them
                         // and the zero row in the others (x + 0 = x: same sums).  Exact-builder kernels only (qk_conv_aprx with lutMode 0)
  int relu;              // fuse max(0, x) into the store
  int panels;
  int lutF16;            // tolerance study (BASELINE configs[4]): table entries rounded to fp16 before they are stored
  // A launch that does not fill the chip (one GPU's share of a sharded batch): the tiles from rank splitFrom on (the
  // tail of the heaviest-first order) are cut into splitZ workgroups each, which take consecutive slices of the tile's
  // stage sequence and leave partial sums in `partial` ([tile - splitFrom][slice][panel][position][Ct][128]); k_conv_sum
  // adds the slices in order (bias sits in slice 0, ReLU is applied there).  splitZ <= 1: no tile is split.  Chosen by
  // qk_conv_plan; the summation order of a split tile differs from the reference's, so the exact builder never splits.
  int splitFrom, splitZ;
  float* partial;
  // Sliding variant (k_conv_aprx<.., SLIDE>, qk_conv_plan_slide): every output column is cut into nSeg segments of
  // output rows [segBeg[i], segBeg[i + 1]) (longest first); a workgroup sweeps one segment.  progS: the program table
  // of that variant ([slots * stride][knl][M][rowU16]).  nSeg = 0: the tile kernel.
  int nSeg;
  int segBeg[9];
  const uint16_t* progS;
};
constexpr int QK_MAX_SEGS = 8;
struct QkSplitPlan {
  int splitFrom;         // first split tile rank (= number of tiles: nothing is split)
  int Z;                 // slices per split tile
  size_t partialFloats;  // scratch the launch needs
  double cost;           // predicted duration of the launch in stage-times (list schedule on 256 CUs + reduction)
};
// decide the split of a conv launch over p.panels panels (p.partial / splitFrom / splitZ are ignored); scratchFloats =
// what the caller can offer for partial sums
QkSplitPlan qk_conv_plan(const ConvParams& p, size_t scratchFloats);
// Segments of the sliding variant for a launch over p.panels panels: fills p.nSeg / p.segBeg when sliding is predicted to
// beat `tileCost` (the list-scheduled stage-times of the tile kernel, QkSplitPlan::cost), else leaves nSeg = 0; returns the
// predicted duration of the chosen sliding launch in stage-times (0: none chosen)
double qk_conv_plan_slide(ConvParams& p, double tileCost);
// Symmetric workgroups (k_conv_sym: all 16 waves build and gather, 8 channels x a 2x2 tile per wave) for layers with exactly
// 128 channels per group: eligibility, predicted duration in stage-times, launch (p.progS = the program table of the
// (8 channels per wave, 2x2 tile) layout: qk_make_slots(128, groups, 8) / qk_conv_program)
bool qk_conv_sym_shape(int Cin, int grp, int Ct, int M, int Cs, int K);
double qk_conv_sym_cost(const ConvParams& p);
hipError_t qk_conv_sym(const ConvParams& p, hipStream_t st);

// Eight-wave symmetric workgroups with 256 registers per wave (qcnn_sym8.hip): all 8 waves build and gather, cpw channels x
// a th x tw tile per wave (cpw * th * tw = 96 = 192 accumulator registers), `chunks` workgroups along the channel axis.
// cpw = 0: the layer is not eligible.  Program table of this layout: [rfH][rfW][M][groups * chunks][8 waves][2 halves]
// [position][cpw / 2] uint16 (ConvParams::progS when the kernel is launched).
struct Qk8Config { int cpw, th, tw, chunks, slide; };   // slide: th = accumulator slots per column, tw = columns of a strip
// position (in floats) inside ConvParams::ctrd8 of code word k (0..127), dim d of sub-space m; ks = Cs / 4 k-steps
__host__ __device__ static inline size_t qk_ctrd8_index(int m, int d, int k, int ks) {
  const int h = k >> 6, i = (k >> 4) & 3, li = k & 15, step = d >> 2, lk = d & 3;
  return ((((size_t)m * 2 + h) * ks + step) * 64 + lk * 16 + li) * 4 + i;
}
Qk8Config qk_conv_sym8_config(int Cin, int grp, int Ct, int M, int Cs, int K);
size_t qk_conv_sym8_program_bytes(const Qk8Config& cf, int groups, int knl, int stride, int M);
hipError_t qk_build_program8(const uint8_t* rows, uint16_t* prog, const QkSlots& src, const Qk8Config& cf, int Ctg, int groups,
                             int knl, int stride, int M, hipStream_t st, int f16 = 0);   // f16: offsets into the fp16 table layout
double qk_conv_sym8_cost(const ConvParams& p, const Qk8Config& cf, double stageFactor, int Z = 1);   // Z: slices per tile (ConvParams::splitZ)
// dst = sum over the Z slices (in slice order) of the partial sums of the tiles from rank splitFrom on; optional ReLU (k_conv_sum)
hipError_t qk_conv_sum(const float* partial, float* dst, int splitFrom, int Z, int panels, int tilesX, int tilesY, int TH, int TW, int Ho,
                       int Wo, int Ct, int relu, hipStream_t st);
hipError_t qk_conv_sym8(const ConvParams& p, hipStream_t st, int mode = 0);   // mode 1: fp16 table storage, 2: + fp16 sums (p.progS built with f16 = 1)
Qk8Config qk_conv_sym8_config16(int Cin, int grp, int Ct, int M, int Cs, int K);   // tiles of mode 2 (twice the positions)
// The sliding form of the eight-wave kernel (k_conv_sym8<.., SLIDE>): config (cpw = 0: not eligible), segments + predicted
// duration, launch.  Program table: qk_conv_sym8_program_bytes / qk_build_program8 with this config.
Qk8Config qk_conv_sym8_slide_config(int Cin, int grp, int Ct, int M, int Cs, int K, int knl, int stride);
double qk_conv_sym8_slide_plan(ConvParams& p, const Qk8Config& cf, double stageFactor);
hipError_t qk_conv_sym8_slide(const ConvParams& p, hipStream_t st);

// Half-panel eight-wave kernel (qcnn_half8.hip, k_conv_half8): workgroups of 64 images — twice the (position, channel) sums per
// workgroup (1536), half the table build per stage.  cpw channels per wave, th x tw tile, ws wave sets that share the channels and
// interleave the positions (cpw * th * tw / ws = 192); cpw = 0: not eligible.  Program table: [rfH][rfW][M][groups * chunks][8 waves]
// [4 lane groups][th * tw / ws][cpw / 4] uint16 (ConvParams::progS when the kernel is launched).
struct QkH8Config { int cpw, th, tw, ws, chunks, slide; };
// a half-panel stage costs QK_HALF8_FIX + QK_HALF8_PER_ROW x (row look-ups of 64 images it serves) cycles (k_conv_sym8: 2540 + 1.97 x
// rows of 128 images); measured: LABBOOK.md, round 6
constexpr double QK_HALF8_FIX = 2040.0, QK_HALF8_PER_ROW = 0.61, QK_HALF8_TWO_SETS = 1.08;
QkH8Config qk_conv_half8_config(int Cin, int grp, int Ct, int M, int Cs, int K);
size_t qk_conv_half8_program_bytes(const QkH8Config& cf, int groups, int knl, int stride, int M);
hipError_t qk_build_program_h8(const uint8_t* rows, uint16_t* prog, const QkSlots& src, const QkH8Config& cf, int Ctg, int groups,
                               int knl, int stride, int M, hipStream_t st);
double qk_conv_half8_cost(const ConvParams& p, const QkH8Config& cf, double stageFactor);
hipError_t qk_conv_half8(const ConvParams& p, hipStream_t st);       // p.progS = the half-panel program, p.ctrd8 as for qk_conv_sym8
// its sliding form (k_conv_half8<.., SLIDE>): config (cpw = 0: not eligible), segments + predicted duration, launch.  Program table:
// qk_conv_half8_program_bytes / qk_build_program_h8 with this config.
QkH8Config qk_conv_half8_slide_config(int Cin, int grp, int Ct, int M, int Cs, int K, int knl, int stride);
double qk_conv_half8_slide_plan(ConvParams& p, const QkH8Config& cf, double stageFactor);
hipError_t qk_conv_half8_slide(const ConvParams& p, hipStream_t st);

struct FcParams {
  float* partial;        // [msplit][panels][Ct][128] scratch for split-M partial sums (msplit > 1)
  int msplit;            // blocks along the sub-space axis (1 = single pass, bit-exact summation order)
  const float* src;      // [panels][D][128]
  float* dst;            // [panels][Ct][128]
  const float* bias;
  const float* ctrd;     // [M][Cs][K]
  const uint8_t* rows;   // [M][rowStride]  (src/CaffeEva.cc:610-611): row slots, QkSlots order
  const uint8_t* cbn;    // the same assignments BIT-PACKED as the reference's .cbn payload holds them (include/FileIO.h:128-166: file
                         // order [Ct][M], 4096-byte blocks of floor(32768 / bits) values, MSB first, 0-based code words), or NULL;
  int cbnBits;           // read in place by the few-image kernel (qk_fc_small): 4 / 5 bits per assignment instead of a byte
  int D, Ct, M, Cs, K;
  int pd;                // as ConvParams::pd
  int relu;
  int panels;
  int lutF16;            // as ConvParams::lutF16
};

// Eight-wave FC kernel (qcnn_sym8.hip, k_fc_sym8): K = 32, Cs = 4, complete sub-spaces; 96 channels per wave, 768 per workgroup.
// prog: [M][chunks][8 waves][2 halves][48] pre-scaled uint16 offsets; ctrdF: the code book in operand order (qk_ctrdf_index).
bool qk_fc_sym8_shape(int D, int Ct, int M, int Cs, int K);
int qk_fc_sym8_chunks(int Ct);
size_t qk_fc_sym8_program_bytes(int Ct, int M);
hipError_t qk_build_program_fc8(const uint8_t* rows, uint16_t* prog, const QkSlots& src, int Ct, int M, hipStream_t st, int f16 = 0);
hipError_t qk_fc_sym8(const FcParams& p, const uint16_t* prog, const float* ctrdF, hipStream_t st, int mode = 0);   // 1: fp16 tables, 2: + fp16 sums
// position (in floats) inside ctrdF of code word k (0..31), dim d (0..3) of sub-space m: stage m / 4, row tile 2 (m % 4) + k / 16
__host__ __device__ static inline size_t qk_ctrdf_index(int m, int d, int k) {
  const int stage = m >> 2, rt = 2 * (m & 3) + (k >> 4), h = rt >> 2, i = rt & 3, li = k & 15;
  return ((((size_t)stage * 2 + h) * 64) + d * 16 + li) * 4 + i;
}

// Precise path (qcnn_dense.hip): conv and FC layers with dense weights (FC = 1x1 conv on a 1x1 map, Cin = D).
struct DenseParams {
  const float* src;      // [panels][H*W*Cin][128]
  float* dst;            // [panels][Ho*Wo*Ct][128]
  const float* bias;     // [Ct]
  const float* wt;       // [grp][kh][kw][Cin/grp][Ct/grp]: the reference's [Ct][Cin/grp][kh][kw] kernels, output channel innermost
  int H, W, Cin, Ho, Wo, Ct;
  int knl, stride, pad, grp;
  int relu;
  int panels;
};
hipError_t qk_dense(const DenseParams& p, hipStream_t st);

// lutMode: 0 exact VALU, 1 f32 MFMA, 2 = f32 MFMA with fp16-rounded table entries kept in f32 slots (see lutF16: the layers the
// fp16-storage kernels of qcnn_sym8.hip do not cover).  Return hipError_t of the launch.
hipError_t qk_conv_aprx(const ConvParams& p, int lutMode, hipStream_t st);
hipError_t qk_fc_aprx(const FcParams& p, int lutMode, hipStream_t st);
int qk_fc_channels_per_block(int Ct);   // output channels one k_fc_aprx workgroup covers
// The same two layers for a batch of a few images (qcnn_small.hip): lanes = output channels, one workgroup per (output
// tile, channel chunk, image); n = images of the launch (all inside the panels src/dst point at); f32 arithmetic,
// results equal to the panel kernels' to rounding.  p.rows / p.ctrd / p.bias as above; msplit / partial unused.
hipError_t qk_conv_small(const ConvParams& p, int n, hipStream_t st);
hipError_t qk_fc_small(const FcParams& p, int n, hipStream_t st);

// Load-time decode of a bit-packed assignment stream (.cbn payload, include/FileIO.h:128-166: 4096-byte blocks of
// floor(32768 / bits) values packed MSB first, 0-based code-word indices in FILE order [Ct][taps][M]) straight
// into the row-offset table [taps][M][rowStride] of the arena.  *bad is set when an index >= K is met.
hipError_t qk_decode_cbn(const uint8_t* blocks, int bits, size_t n, int Ct, int taps, int M, int K, QkSlots sl,
                         uint8_t* rows, int* bad, hipStream_t st);

// rows (plain table of a conv layer) -> prog (QkProgram order); one thread per program entry
// rows: the plain table in the order of `src` slots; prog: program in the order of `dst` slots (the same, or the sliding
// variant's own channel split)
hipError_t qk_build_program(const uint8_t* rows, uint16_t* prog, QkSlots src, QkSlots dst, QkProgram pg, int knl, int stride,
                            int M, hipStream_t st, int slide = 0);

// dst row e = src row map[e], rows of 128 images ([panels][D][128]); the first FC layer consumes its input
// NCHW-flattened (src/CaffeEva.cc:187-189)
// `live` (here and in the glue wrappers below): images every panel of the launch really holds — 128, or the batch size
// of a single-panel launch, whose other lanes are then not touched at all
hipError_t qk_permute_rows(const float* src, float* dst, const int* map, int D, int panels, int live, hipStream_t st);
// dst[e] = sum_z partial[z][e] (z ascending), optional ReLU; n floats per partial slab
hipError_t qk_sum_partials(const float* partial, float* dst, int msplit, size_t n, int relu, hipStream_t st);

hipError_t qk_relu(const float* src, float* dst, size_t n, hipStream_t st);
hipError_t qk_lrn(const float* src, float* dst, int panels, int HW, int C, int lrnSiz, float alp, float bet,
                  float ini, int live, hipStream_t st);
// LRN and the 3x3 / stride 2 / pad 0 max-pool behind it in one pass (the normalised map is not written)
int qk_lrn_pool_blocks(int Ho, int Wo);          // workgroups per panel
hipError_t qk_lrn_pool(const float* src, float* dst, int panels, int H, int W, int C, int Ho, int Wo, int lrnSiz, float alp,
                       float bet, float ini, int live, hipStream_t st);
hipError_t qk_pool(const float* src, float* dst, int panels, int H, int W, int C, int Ho, int Wo, int knl,
                   int stride, int pad, int live, hipStream_t st);
// Quantised conv layer with one sub-space of <= 4 dims evaluated through its decoded code words on the matrix pipe
// (qcnn_decoded.hip).  wdec: [knl][Kp][S] — kernel row, k = column * Cin + d (padded to Kp, zeros), channel (row stride S)
struct DecParams {
  int srcNchw;           // 1: src is the network input [nImages][Cin][H][W] read in place (qk_conv_dec_nchw); 0: panels
  int nImages, panel0;   // srcNchw: images of the batch, index of the launch's first panel inside it
  const float* src;      // [panels][H*W*Cin][128]
  float* dst;            // [panels][Ho*Wo*Ct][128]
  const float* bias;     // [Ct]
  const float* wdec;
  int H, W, Cin, Ho, Wo, Ct;
  int knl, stride, pad;
  int Kr, Kp, S;         // knl * Cin, the same rounded up to a multiple of 4, channel stride of wdec
  int relu, panels, live;
};
bool qk_conv_dec_shape(int Cin, int grp, int M, int Ct, int knl, int* Kp, int* S);   // is the layer eligible (and its wdec shape)
hipError_t qk_decode_weights(const uint8_t* rows, const float* ctrd, float* out, const QkSlots& sl, int knl, int Cin, int K,
                             int Ct, int Kp, int S, hipStream_t st);
hipError_t qk_conv_dec(const DecParams& p, hipStream_t st);
// The same layer reading the NCHW network input in place (no pack pass), k flat over the window (Cin knl^2 products in
// fours, padded to Kp = a multiple of 16): unpadded layers.  wdec: [Kp / 4 steps][S / 16][4 k][16], S = Ct.
bool qk_conv_dec_nchw_shape(int Cin, int grp, int M, int Ct, int knl, int pad, int* Kp, int* S);
hipError_t qk_decode_weights_nchw(const uint8_t* rows, const float* ctrd, float* out, const QkSlots& sl, int knl, int Cin, int K,
                                  int Ct, int Kp, int S, hipStream_t st);
hipError_t qk_conv_dec_nchw(const DecParams& p, hipStream_t st);   // p.Kr = Cin knl^2, p.Kp and p.S from qk_conv_dec_nchw_shape
// FC layer with one-dim sub-spaces through its decoded code words (qcnn_decoded.hip).  wdec: [D][S], S = Ct rounded up to 64
struct FcDecParams {
  const float* src;      // [panels][D][128]
  float* dst;            // [panels][Ct][128]
  float* partial;        // [slices][panels][Ct][128] when the k axis is cut over workgroups (qk_fc_dec_slices > 1)
  const float* bias;
  const float* wdec;
  int D, Ct, S;
  int relu, panels, halves;
};
bool qk_fc_dec_shape(int D, int M, int Cs, int Ct, int* S);
hipError_t qk_decode_fc_weights(const uint8_t* rows, const float* ctrd, float* out, const QkSlots& sl, int D, int K, int Ct,
                                int S, hipStream_t st);
int qk_fc_dec_slices(int D, int Ct, int panels, int live);
hipError_t qk_fc_dec(const FcDecParams& p, int slices, int live, hipStream_t st);   // slices > 1: add with qk_sum_partials
hipError_t qk_softmax(const float* src, float* dst, int panels, int C, int live, hipStream_t st);
hipError_t qk_top5(const float* prob, uint16_t* out, int n, int C, hipStream_t st);   // prob panel layout -> [n][5]

// [n][C][H][W] -> panels [H*W*C][128] (lanes >= n zero-filled)
hipError_t qk_pack_nchw(const float* in, float* dst, int n, int C, int H, int W, hipStream_t st);
// 8-bit planar [n][C][Hs][Ws] minus mean [C][Hs][Ws] (or NULL), centre crop H x W -> panels [H*W*C][128]
hipError_t qk_pack_u8(const uint8_t* in, const float* mean, float* dst, int n, int C, int H, int W, int Hs, int Ws,
                      hipStream_t st);
// [n][E] (already in NHWC / flat order) -> panels [E][128]
hipError_t qk_pack_rows(const float* in, float* dst, int n, int E, hipStream_t st);
// panels [E][128] -> [n][E]
hipError_t qk_unpack_rows(const float* src, float* out, int n, int E, hipStream_t st);

#endif  // QCNN_KERNELS_H_
