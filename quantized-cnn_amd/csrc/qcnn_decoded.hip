// qcnn_decoded.hip — quantised conv layers with ONE sub-space of <= 4 dims (AlexNet conv1, VGG-16 conv1_1: the RGB input).
//
// For such a layer a table look-up replaces Cin <= 4 multiply-adds and the table of a source pixel (K code words x 128
// images, 64 KB of LDS stores) serves only knl^2 / stride^2 x Ct look-ups: in k_conv_aprx two thirds of a conv1 stage are
// the table build.  The same sum
//
//     dst[pos][c] = bias[c] + sum_taps  LUT[pixel(pos, tap)][asmt[tap][c]],     LUT[p][k] = sum_d x[p][d] * ctrd[d][k]
//                                                                   (src/CaffeEva.cc:816-865 over :1261-1296)
//
// is evaluated here WITHOUT materialising the tables: every assignment is replaced by the code word it names
// (k_decode_weights: w[tap][d][c] = ctrd[d][asmt[tap][c]], once per parameter upload) and the products go to the matrix
// pipe — v_mfma_f32_16x16x4_f32 with A = decoded code words [16 channels x 4 k], B = the panel rows themselves
// [4 k x 16 images]; k runs over the (column, input channel) pairs of ONE kernel row, which are consecutive panel rows,
// so there is no im2col buffer and no address table.  Same quantised parameters, same function; the sum is fused
// multiply-adds in (tap, d) order instead of d-sums rounded into a table first — within the north-star tolerance like
// every MFMA path (the exact builder, QCNN_LUT_EXACT, never takes this path).
//
// Kernels: k_conv_dec (below: persistent waves, all code words of the layer in LDS, wide buffer loads) and, for FC layers
// whose sub-spaces have ONE dim, k_fc_dec (further down).  Both are opt-out (QCNN_OPT_DECODE = 0: table kernels).
#include "qcnn_kernels.h"
#include <algorithm>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef QCNN_DEC_VAR
#define QCNN_DEC_VAR 0      // timing experiments only (scripts/build_variant.sh, scripts/variants_dec.sh): 1 no B loads, 2 no A reads, 8 no stores
#endif

namespace {

constexpr int PANEL = QCNN_PANEL;

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Product index of padded row k (< Kp) of a kernel row with Kr = knl * Cin real products: the steps (four k each) cover
// [0, 4), [4, 8), ... and the LAST one [Kr - 4, Kr) (Kr >= 4: it overlaps the step before it; an overlapped row carries a
// zero code word in the last step, reported as -1) — every operand row a step loads lies inside the window.  Kr < 4: the
// single step is [0, 4), rows >= Kr are padding (-1; the kernel clamps their loads: PADDED path).
__host__ __device__ __forceinline__ int qk_dec_krow(int k, int Kr, int Kp) {
  const int last = Kp - 4;                                  // first padded row of the last step
  if (k < last || Kr < 4) return k < Kr ? k : -1;
  const int real = Kr - 4 + (k - last);
  return real >= last ? real : -1;
}

// rows: [kh][kw][M = 1][rowStride] slot bytes (QkSlots order); ctrd: [Cs][K]; out: [knl][Kp][S]
__global__ void k_decode_weights(const uint8_t* __restrict__ rows, const float* __restrict__ ctrd, float* __restrict__ out,
                                 QkSlots sl, int knl, int Cin, int K, int Ct, int Kp, int S) {
  const int total = knl * Kp * S;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ch = i % S, k = qk_dec_krow((i / S) % Kp, knl * Cin, Kp), kh = i / (S * Kp);
    float w = 0.0f;
    if (ch < Ct && k >= 0) {
      const int kw = k / Cin, d = k % Cin;
      const int slot = rows[(size_t)(kh * knl + kw) * sl.rowStride + qk_slot_entry(sl, 0, ch)];
      w = ctrd[(size_t)d * K + qcnn_row_slot(slot)];            // qcnn_row_slot is its own inverse; M = 1: stage row = code word
    }
    out[i] = w;
  }
}

// Persistent waves: a workgroup loads ALL decoded code words into LDS once ([knl * Kp][S], <= 152 KB), then every wave
// walks its own list of work items without any workgroup synchronisation: the waves of a SIMD drift apart, so one wave's
// operand loads and result stores sit under the other waves' matrix instructions.
//
// A work item = (panel, PW consecutive output positions, 64 of the panel's images, 16 * CT channels).  Every memory
// instruction between two matrix instructions costs matrix-pipe time (measured: ~20 cycles per load, ~55 per store), so
// the operands come in wide: ONE buffer_load_dwordx4 brings the B operands of FOUR image tiles — a lane (k row kq, li)
// holds images 4 li .. 4 li + 3 of its 64, image tile t = the images with index t modulo 4, any partition of the images
// into sixteens serves —, one ds_read2_b32 the A operands of two channel tiles; the results leave as dwordx4 rows of four
// consecutive images.  B operands run three steps ahead of the products (a ring of three register sets over the flat
// (kernel row, step) sequence), A operands one step ahead.
// IT = image tiles per wave: 4 (64 images, dwordx4 operands and results) or 1 (16 images, dwords: batches of <= 16 images,
// where three quarters of a 64-image item would multiply zeros).
template <int CT, int PW, bool PADDED, int R, int IT>
__global__ __launch_bounds__(1024) void k_conv_dec(DecParams p) {
  extern __shared__ __attribute__((aligned(16))) float ldsW[];          // [knl * Kp][S]
  const int lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
  const int P = p.Ho * p.Wo;
  {
    const int wq = (p.knl * p.Kp * p.S) >> 2;
    const f32x4* __restrict__ wsrc = reinterpret_cast<const f32x4*>(p.wdec);
    f32x4* ldsW4 = reinterpret_cast<f32x4*>(ldsW);
    for (int i = threadIdx.x; i < wq; i += 1024) ldsW4[i] = wsrc[i];
  }
  __syncthreads();
  const int NS = p.Kp >> 2;                                             // steps (of four k) per kernel row
  const int T = p.knl * NS;                                             // steps per work item
  const int groups = (P + PW - 1) / PW;                                 // position groups per panel
  constexpr int IB = 16 * IT;                                           // images per work item
  const int halves = (p.live + IB - 1) / IB;                            // image blocks a panel has (a small batch: fewer)
  const int chunks = p.Ct / (16 * CT);                                  // channel chunks
  const int nItems = p.panels * groups * halves * chunks;
  const int kClamp = p.Kr - 1;
  // Workgroups go to the eight XCDs round-robin (workgroup i -> XCD i % 8) and every XCD has its own L2: an XCD takes a
  // CONTIGUOUS eighth of the item list (whole panels for a 1000-image batch), so that the windows its waves read overlap
  // in ITS L2 instead of every L2 fetching the whole input.
  const int xcd = blockIdx.x & 7, nX = gridDim.x < 8 ? gridDim.x : 8;
  const int wgX = (gridDim.x - xcd + 7) >> 3;                           // workgroups of this XCD
  const int itemBeg = (int)((long long)nItems * xcd / nX), itemEnd = (int)((long long)nItems * (xcd + 1) / nX);
  for (int item = itemBeg + (blockIdx.x >> 3) * 16 + wave; item < itemEnd; item += wgX * 16) {
    // The lane-derived address parts are RE-DERIVED per item from an opaque copy of the lane id: as loop invariants they
    // would have to live through the whole item loop beside 96 accumulator registers, and the compiler spilled two of
    // them to scratch (a handful of vector instructions per ~76 000-cycle item instead).
    int laneI = lane;
    asm volatile("" : "+v"(laneI));
    const int li = laneI & 15, kq = laneI >> 4;
    // channel chunk fastest, then image block: the waves of a workgroup share their B rows through the L1
    const int cc = item % chunks, ib = (item / chunks) % halves, pgi = item / (chunks * halves);
    const int panel = pgi / groups, pos0 = (pgi % groups) * PW;
    int r0[PW], c0[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int q = pos0 + j < P ? pos0 + j : P - 1;                    // positions past the map (last group): computed, not stored
      r0[j] = (q / p.Wo) * p.stride - p.pad;
      c0[j] = (q % p.Wo) * p.stride - p.pad;
    }
    // Vector ALU work takes matrix-pipe time on a SIMD (LABBOOK.md), so a B load costs no vector instruction besides
    // itself: a buffer load with the lane's part of the address (k row kq, images 4 li ..) in ONE constant VGPR and
    // everything else — position, kernel row, step — in the scalar offset.  A kernel row of Kr = knl * Cin products is
    // padded to Kp = a multiple of four by letting its LAST step start at k = Kr - 4 (qk_dec_krow): it re-reads rows the
    // step before it already multiplied — their code words are zero in that step (k_decode_weights) — instead of rows of
    // the pixel next to the window: 0 x Inf / NaN of a value outside the window must not reach an output the window does
    // not cover (the table kernels have no such coupling either).  All in the scalar offset: no vector instruction.
    const float* __restrict__ srcU = p.src + (size_t)panel * p.H * p.W * p.Cin * PANEL + ib * IB;
    const size_t left = (size_t)(p.panels - panel) * p.H * p.W * p.Cin * PANEL * sizeof(float) - ib * IB * sizeof(float);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(srcU), 0, left < 0xffffffffull ? (unsigned)left : 0xffffffffu, 0x00020000);
    const int laneOff = (kq * PANEL + IT * li) * (int)sizeof(float);
    int posOff[PW];                                                     // byte offset of the window's first row, per position
#pragma unroll
    for (int j = 0; j < PW; ++j) posOff[j] = (r0[j] * p.W + c0[j]) * p.Cin * PANEL * (int)sizeof(float);
    const int rowPitch = p.W * p.Cin * PANEL * (int)sizeof(float);
    constexpr int kStep = 4 * PANEL * (int)sizeof(float);                // bytes between two steps' first rows
    // B operands of step `step` of kernel row `kh`: four consecutive panel rows (k) x 64 images, per position
    auto load_b = [&](int kh, int step, int soff, float (&b)[PW][IT]) {
#if QCNN_DEC_VAR & 1
      for (int j = 0; j < PW; ++j) for (int ti = 0; ti < IT; ++ti) b[j][ti] = (float)(kh + step + j + ti);
      return;
#endif
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        if (PADDED) {
          const int row = r0[j] + kh;
          const int k = (step == NS - 1 && p.Kr >= 4) ? p.Kr - 4 + kq : 4 * step + kq;   // the last step ends at the window's last row
          const int kc = k < kClamp ? k : kClamp;                       // k >= Kr (Kr < 4): an operand of the window, its code word is zero
          const int col = c0[j] + kc / p.Cin;
          const bool ok = row >= 0 && row < p.H && col >= 0 && col < p.W;
          const float* __restrict__ px = srcU + (ok ? ((row * p.W + c0[j]) * p.Cin + kc) * PANEL + IT * li : 0);   // a panel map is < 2^31 floats (qk_conv_dec)
          if (IT == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(px);
#pragma unroll
            for (int ti = 0; ti < IT; ++ti) b[j][ti] = ok ? v[ti] : 0.0f;
          } else {
            const float v = *px;
            b[j][0] = ok ? v : 0.0f;
          }
        } else if (IT == 4) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, laneOff, posOff[j] + soff, 0));
#pragma unroll
          for (int ti = 0; ti < IT; ++ti) b[j][ti] = v[ti];
        } else {
          b[j][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, laneOff, posOff[j] + soff, 0));
        }
      }
    };
    f32x4 acc[PW][CT][IT];                                              // [position][channel tile][image tile]
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + (cc * CT + ct) * 16 + 4 * kq);   // D row 4 * kq + r = channel
#pragma unroll
      for (int j = 0; j < PW; ++j)
#pragma unroll
        for (int ti = 0; ti < IT; ++ti) acc[j][ct][ti] = b4;
    }
    // (kernel row, step) of the next B load and its byte offset from the window's first row.  Past the last step the
    // sequence simply runs on (rows below the window, or the buffer's zero): those operands are never multiplied, and
    // loads that are unconditional let the compiler count the ones in flight.
    const int lastOff = NS > 1 ? (p.Kr - 4) * PANEL * (int)sizeof(float) : 0;   // first row of a kernel row's last step
    const int incLast = NS > 1 ? lastOff - (NS - 2) * kStep : 0;        // from the step before the last one to the last one
    const int incWrap = rowPitch - lastOff;                             // from a row's last step to the next row's first
    int khL = 0, stL = 0, soffL = 0;
    float b[R][PW][IT];                                                 // R = operand sets in flight (2 or 3)
    float a[R][CT];
    auto next_b = [&](float (&bb)[PW][IT]) {
      load_b(khL, stL, soffL, bb);
      const int wrap = stL + 1 == NS ? 1 : 0;
      stL = wrap ? 0 : stL + 1;
      khL += wrap;
      soffL += wrap ? incWrap : (stL == NS - 1 ? incLast : kStep);
    };
    // A operands of flat step t: k rows 4 t .. 4 t + 3 (= kh * Kp + 4 * step) of this wave's channels; past the end: LDS zeros
    const float* __restrict__ wl = ldsW + kq * p.S + cc * 16 * CT + li;
    const int aStep = 4 * p.S;
    auto load_a = [&](const float* __restrict__ w, float (&aa)[CT]) {
#if QCNN_DEC_VAR & 2
      for (int ct = 0; ct < CT; ++ct) aa[ct] = (float)ct;
      return;
#endif
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) aa[ct] = w[ct * 16];
    };
    auto products = [&](const float (&aa)[CT], const float (&bb)[PW][IT]) {
#pragma unroll
      for (int j = 0; j < PW; ++j)
#pragma unroll
        for (int ti = 0; ti < IT; ++ti)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[j][ct][ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ct], bb[j][ti], acc[j][ct][ti], 0, 0, 0);
    };
#pragma unroll
    for (int u = 0; u < R; ++u) next_b(b[u]);
    load_a(wl, a[0]);
    int t = 0;
    for (; t + R <= T; t += R) {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        wl += aStep;
        load_a(wl, a[(u + 1) % R]);
        __builtin_amdgcn_sched_barrier(0);
        products(a[u], b[u]);
        __builtin_amdgcn_sched_barrier(0);
        next_b(b[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < R - 1; ++u)                                     // the steps left when T is not a multiple of R
      if (t + u < T) {
        wl += aStep;
        load_a(wl, a[(u + 1) % R]);
        products(a[u], b[u]);
      }
    // D tile ti, element r of lane (li, kq): channel 4 * kq + r of the tile, image IT * li + ti of the block
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      if (pos0 + j >= P) continue;
      float* __restrict__ dst = p.dst + ((size_t)panel * P + pos0 + j) * p.Ct * PANEL + ib * IB + IT * li;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v[IT];
#pragma unroll
          for (int ti = 0; ti < IT; ++ti) {
            v[ti] = acc[j][ct][ti][r];
            if (p.relu) v[ti] = (0.0f < v[ti]) ? v[ti] : 0.0f;
          }
          float* __restrict__ o = dst + (size_t)((cc * CT + ct) * 16 + 4 * kq + r) * PANEL;
#if QCNN_DEC_VAR & 8
          if (v[0] == 1.2345f)
#endif
          {
            if (IT == 4) *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[IT > 1 ? 1 : 0], v[IT > 2 ? 2 : 0], v[IT > 3 ? 3 : 0]};
            else *o = v[0];
          }
        }
    }
  }
}

template <int CT, int PW, bool PADDED, int R, int IT>
hipError_t launch_dec(const DecParams& p, hipStream_t st) {
  const int P = p.Ho * p.Wo;
  const long long items = (long long)p.panels * ((P + PW - 1) / PW) * ((p.live + 16 * IT - 1) / (16 * IT)) * (p.Ct / (16 * CT));
  const int blocks = (int)std::min<long long>(256, (items + 15) / 16);    // one persistent workgroup per CU
  const size_t shm = (size_t)p.knl * p.Kp * p.S * sizeof(float);
  auto kern = k_conv_dec<CT, PW, PADDED, R, IT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), shm, st, p);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// k_conv_dec_nchw: the same products with the operands taken straight from the NCHW network input — no pack pass (0.24 ms
// per 1000 AlexNet images: 1.24 GB through HBM) in front of the first layer, and k FLAT over the window: k = (c knl + kh) knl
// + kw runs 0 .. Cin knl^2 - 1 in fours whatever the length of a kernel row, so the matrix pipe multiplies 364 (368 with the
// steps padded to a multiple of four; not 11 x 36 = 396) products per output for AlexNet conv1.  Lane (k row kq, li) of step
// s needs element k = 4 s + kq of a window: byte offset c H W 4 + kh W 4 + kw 4 from the window's corner — a table of steps
// x 4 ints the workgroup computes once into LDS — plus the lane's image and position; the item's position group, image tile
// and panel travel in the scalar offset.
// Work item of a wave: 16 images x FOUR NEIGHBOURING POSITIONS of an output row x 96 channels (24 accumulator tiles).  In
// NCHW the images are C H W floats apart, so the 16 rows of a product tile are 2 positions x 8 images: with stride 4 the
// lanes of one image read 4 k x 2 positions = 8 consecutive floats, a dword load touches 8 lines of 32 useful bytes, and the
// four loads of a step and the steps of a kernel row hit the same lines again.  The image values are the A operand (rows of
// the product), the code words B: a lane ends up with FOUR CONSECUTIVE IMAGES of one channel — 24 16-byte stores per item.
// Eight waves of up to 256 registers per CU, a ring of four steps' operands (three in flight, counted s_waitcnt).
// Measured, AlexNet conv1 at 1000 images (scripts/variants_nchw.sh; pack + k_conv_dec: 0.24 + 1.76 ms):
//   tile = 16 images x 1 position, item = 64 images (dword stores)      2.34 ms   (L1: 8 waves x 64 lines per kernel row)
//   tile = 16 images x 1 position, item = 16 images x 4 positions       1.98 ms, 16-byte stores 1.86 ms (loads alone 1.37 ms)
//   tile = 4 images x 4 positions                                       1.96 ms   (loads alone 0.67 ms, 16-byte store fragments)
//   tile = 8 images x 2 positions                                       1.85 ms   (everything but the products 1.10 ms,
//                                                                                  everything but loads or but stores 1.68 ms)
//   one loop body (no peeled tail, no zero-trip path: 225 -> 133 registers), then 5 / 6 positions of 16 images per item
//   1.86 / 2.06 ms, 12 / 16 waves per workgroup 1.86 / 1.88 ms against 1.81 (synthetic parameters): eight waves, four positions;
//   the two 32-byte halves of a row stored back to back 1.79 ms; non-temporal stores 1.86; item = 32 images x 2 positions
//   (whole 128-byte lines per wave) 1.82
// the matrix pipe alone would need 1.42 ms at 2.4 GHz; what is left is the stores and loads of a wave's item boundary that
// its SIMD neighbour's products do not cover.
// ------------------------------------------------------------------------------------------------------------------
// s_waitcnt vmcnt(N) that the operands of a step pass through (the compiler must not move their uses above it)
template <int N, int IT>
__device__ __forceinline__ void nchw_wait_impl(float (&bb)[IT]) {
  if constexpr (IT == 4) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(bb[0]), "+v"(bb[1]), "+v"(bb[2]), "+v"(bb[3]) : "n"(N));
  else if constexpr (IT == 5) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(bb[0]), "+v"(bb[1]), "+v"(bb[2]), "+v"(bb[3]), "+v"(bb[4]) : "n"(N));
  else asm volatile("s_waitcnt vmcnt(%6)" : "+v"(bb[0]), "+v"(bb[1]), "+v"(bb[2]), "+v"(bb[3]), "+v"(bb[4]), "+v"(bb[5]) : "n"(N));
}
template <int N, int IT>
__device__ __forceinline__ void nchw_wait(float (&bb)[IT]) { nchw_wait_impl<N, IT>(bb); }

#ifndef NCHW_WAVES
#define NCHW_WAVES 8
#endif
template <int CT, int IT>
__global__ __launch_bounds__(64 * NCHW_WAVES) void k_conv_dec_nchw(DecParams p) {
  static_assert(IT == 4, "a product tile is 2 positions x 8 images, an item two pairs of positions x two image halves");
  extern __shared__ __attribute__((aligned(16))) float ldsW[];          // [steps][Ct / 16][4 k][16]: a wave's read of one channel tile is
                                                                        // 64 consecutive floats (no bank conflicts); then int [steps + 4][4]
  const int lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
  const int P = p.Ho * p.Wo;
  const int steps = p.Kp >> 2;                                          // Kp: Cin knl^2 padded to a multiple of 16
  const uint32_t imgBytes = (uint32_t)p.Cin * p.H * p.W * 4u, planeBytes = (uint32_t)p.H * p.W * 4u, rowBytes = (uint32_t)p.W * 4u;
  // (one step of slack behind the code words: the last step of an item pre-loads "the next step's" code words, which nobody
  // uses — the offset table sits behind that slack, inside the allocation)
  int* ldsOff = reinterpret_cast<int*>(ldsW + (size_t)(steps + 1) * 4 * p.S);
  {
    const int wq = steps * p.S;
    const f32x4* __restrict__ wsrc = reinterpret_cast<const f32x4*>(p.wdec);
    f32x4* ldsW4 = reinterpret_cast<f32x4*>(ldsW);
    for (int i = threadIdx.x; i < wq; i += 64 * NCHW_WAVES) ldsW4[i] = wsrc[i];
    for (int i = threadIdx.x; i < (steps + 4) * 4; i += 64 * NCHW_WAVES) {
      const int k = min(i, p.Kr - 1);                                   // Kr: Cin knl^2
      const int kw = k % p.knl, kh = (k / p.knl) % p.knl, c = k / (p.knl * p.knl);
      ldsOff[i] = (int)((uint32_t)c * planeBytes + (uint32_t)kh * rowBytes + (uint32_t)kw * 4u);
    }
  }
  __syncthreads();
  const int tiles = (p.live + 15) / 16;                                 // image tiles a panel has (a small batch: fewer)
  const int chunks = p.Ct / (16 * CT);
  const int WoG = (p.Wo + IT - 1) / IT, PG = p.Ho * WoG;                // position groups: IT consecutive positions of an output row
  const int nItems = p.panels * PG * tiles * chunks;
  const unsigned long long total = (unsigned long long)p.nImages * imgBytes;
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  const unsigned long long srcA = reinterpret_cast<unsigned long long>(p.src);
  // buffer resource: base, stride 0, the batch's bytes, raw 32-bit data format.  No load below depends on the range check:
  // every address is inside the batch by construction (see `edge`)
  const i32x4_t rsrc4 = {(int)(unsigned)srcA, (int)((unsigned)(srcA >> 32) & 0xffffu),
                         (int)(total < 0xffffffffull ? (unsigned)total : 0xffffffffu), 0x00020000};
  const int xcd = blockIdx.x & 7, nX = gridDim.x < 8 ? gridDim.x : 8;
  const int wgX = (gridDim.x - xcd + 7) >> 3;
  const int itemBeg = (int)((long long)nItems * xcd / nX), itemEnd = (int)((long long)nItems * (xcd + 1) / nX);
  // a launch of less than one item per wave (a few images) spreads its items over the SIMDs of ALL workgroups — wave w of
  // workgroup g takes item w wgX + g — instead of filling the eight waves of a few: an item then has a matrix pipe to itself
  const bool sparse = itemEnd - itemBeg <= wgX * NCHW_WAVES;
  for (int item = itemBeg + (sparse ? wave * wgX + (int)(blockIdx.x >> 3) : (int)(blockIdx.x >> 3) * NCHW_WAVES + wave); item < itemEnd;
       item += wgX * NCHW_WAVES) {
    int laneI = lane;
    asm volatile("" : "+v"(laneI));                                     // lane-derived constants re-derived per item (registers)
    const int li = laneI & 15, kq = laneI >> 4;
    // image tiles fastest: the eight waves of a workgroup write the 128 images of the same (position, channel) rows at about
    // the same time, so that L2 sees whole lines (position groups fastest: 2.03 against 1.98 ms with dword stores)
    const int it = item % tiles, cc = (item / tiles) % chunks, pg = (item / (tiles * chunks)) % PG, panel = item / (chunks * tiles * PG);
    const int orow = pg / WoG, ocol = (pg % WoG) * IT;
    const int r0 = orow * p.stride, c0 = ocol * p.stride;                        // unpadded layers
    const uint32_t img0 = (uint32_t)(p.panel0 + panel) * PANEL + (uint32_t)it * 16u;
    // row li of a product tile: image li & 7 of the tile's eight, position li >> 3 of its two.  Positions past the end of the
    // output row read columns of the next image row, of the next plane, of the next image ... finite values, never stored.
    // That stays inside the caller's buffer as long as an image FOLLOWS the tile's sixteen.  An item that holds the batch's
    // last image or images past it (a ragged last panel launches all eight image tiles) takes the `edge` form of the body:
    // every image index is clamped to the last image and every position to the last of its output row — in the scalar offset
    // AND per lane —, so that no address leaves the batch whatever the buffer's range check does with the scalar offset (the
    // gfx9 raw-buffer check covers the vector offset only), and (last image) x imgBytes cannot wrap 32 bits.  Those lanes'
    // results are never stored / never read.
    const bool edge = img0 + 16u >= (uint32_t)p.nImages;
    const int laneOff = (int)((uint32_t)(li & 7) * imgBytes + (uint32_t)((li >> 3) * p.stride) * 4u);
    const uint32_t base0 = img0 * imgBytes + (uint32_t)(r0 * p.W + c0) * 4u;      // tile ti: + 8 (ti & 1) images, + 2 (ti >> 1) positions
    const int* __restrict__ offT = ldsOff + kq;
   auto body = [&](auto edgeTag) {
    constexpr bool EDGE = decltype(edgeTag)::value;
    int laneOffE[IT];
    uint32_t baseE[IT];
    if constexpr (EDGE) {
      const uint32_t lastImg = (uint32_t)p.nImages - 1u, lastPos = (uint32_t)p.Wo - 1u;
#pragma unroll
      for (int ti = 0; ti < IT; ++ti) {
        const uint32_t imgS = min(img0 + 8u * (uint32_t)(ti & 1), lastImg), imgL = min(img0 + 8u * (uint32_t)(ti & 1) + (uint32_t)(li & 7), lastImg);
        const uint32_t posS = min((uint32_t)ocol + 2u * (uint32_t)(ti >> 1), lastPos), posL = min((uint32_t)ocol + 2u * (uint32_t)(ti >> 1) + (uint32_t)(li >> 3), lastPos);
        laneOffE[ti] = (int)((imgL - imgS) * imgBytes + (posL - posS) * (uint32_t)p.stride * 4u);
        baseE[ti] = imgS * imgBytes + ((uint32_t)(r0 * p.W) + posS * (uint32_t)p.stride) * 4u;
      }
    }
    // Operand loads as inline assembly with counted waits: the compiler's own vmcnt bookkeeping drains the ring to one
    // step at every loop back edge (s_waitcnt vmcnt(4) in front of the first of four steps).  Loads return in order, so
    // "at most 12 outstanding" = everything but the three newest steps has arrived — whatever else (the previous item's
    // stores) is still in flight only makes the wait stricter.
#ifndef NCHW_VAR
#define NCHW_VAR 0                      // timing experiments (scripts/variants_nchw.sh): 1 no operand loads, 2 no products, 4 no stores
#endif
    auto issue = [&, rsrc4, base0](int s, float (&bb)[IT]) {             // the operands of step s (explicit captures: asm operands inside a generic lambda)
      const int ot = offT[s * 4];
      const int vo = ot + laneOff;
      (void)base0;
#pragma unroll
      for (int ti = 0; ti < IT; ++ti)
        if (NCHW_VAR & 1) asm volatile("v_mov_b32 %0, %1" : "=v"(bb[ti]) : "v"(vo)); else if constexpr (EDGE)
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(bb[ti]) : "v"(ot + laneOffE[ti]), "s"(rsrc4), "s"(baseE[ti])); else
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(bb[ti]) : "v"(vo), "s"(rsrc4),
                     "s"(base0 + (uint32_t)(ti & 1) * 8u * imgBytes + (uint32_t)((ti >> 1) * 2 * p.stride) * 4u));
    };
    f32x4 acc[CT][IT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float b1 = p.bias[(cc * CT + ct) * 16 + li];
#pragma unroll
      for (int ti = 0; ti < IT; ++ti) acc[ct][ti] = f32x4{b1, b1, b1, b1};
    }
    const float* __restrict__ wl = ldsW + cc * 64 * CT + laneI;
    const int aStep = 4 * p.S;
    float a[2][CT], b[4][IT];                                           // b: a ring of four steps' operands, three in flight (a ring
                                                                        // of eight: no faster)
    auto load_a = [&](const float* __restrict__ w, float (&aa)[CT]) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) aa[ct] = w[ct * 64];
    };
    issue(0, b[0]);
    issue(1, b[1]);
    issue(2, b[2]);
    load_a(wl, a[0]);
#define NCHW_WAIT(n, bb) nchw_wait<n * IT>(bb)
#define NCHW_STEP(u, n) /* step s + u; the loads of at most n steps may still be in flight */                                       \
  {                                                                                                                   \
    wl += aStep;                                                                                                      \
    load_a(wl, a[((u) + 1) & 1]); /* code words of the next step */                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    NCHW_WAIT(n, b[u]);                                                                                               \
    _Pragma("unroll") for (int ti = 0; ti < IT; ++ti)                                                                 \
      _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                                               \
        if (NCHW_VAR & 2) asm volatile("" : "+v"(acc[ct][ti]) : "v"(a[(u) & 1][ct]), "v"(b[u][ti])); else             \
        acc[ct][ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[u][ti], a[(u) & 1][ct], acc[ct][ti], 0, 0, 0);           \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  }
    // three steps ahead, into the set the step before released.  The last iteration fetches three steps past the end (the
    // offset table repeats the window's last element there): 12 loads of 4 steps x IT per item that nobody reads — the
    // price of ONE loop body: a peeled tail with decreasing wait counts made the compiler keep the accumulators in a
    // second register block (225 registers; 5 positions per item did not fit)
    int s = 0;
    do {                                                                // (at least four steps: no zero-trip copy of the accumulators)
      issue(s + 3, b[3]); NCHW_STEP(0, 3)
      issue(s + 4, b[0]); NCHW_STEP(1, 3)
      issue(s + 5, b[1]); NCHW_STEP(2, 3)
      issue(s + 6, b[2]); NCHW_STEP(3, 3)
    } while ((s += 4) < steps);
    NCHW_WAIT(0, b[0]); NCHW_WAIT(0, b[1]); NCHW_WAIT(0, b[2]);         // nothing in flight into registers the stores may reuse
#undef NCHW_STEP
#undef NCHW_WAIT
    // The image values are the A operand, the code words B (the lane layouts of the two operands of a 16x16x4 instruction
    // are the same: this is the order of the arguments only): lane (li, kq) holds channel li of the tile at position kq >> 1
    // of the tile's two for the FOUR CONSECUTIVE images 4 (kq & 1) .. + 3 of its eight — one 16-byte store per tile
    {
      const int nPos = (NCHW_VAR & 4) ? (p.Wo < 0 ? IT : 0) : min(IT, p.Wo - ocol);
      float* __restrict__ dst = p.dst + (((size_t)panel * P + orow * p.Wo + ocol + (kq >> 1)) * p.Ct + li) * PANEL + it * 16 + 4 * (kq & 1);
      // two copies of the store loop under one uniform branch: a ReLU applied under a branch INSIDE the loop made the compiler
      // keep a second set of result registers (225 instead of 140)
      if (p.relu) {
        // the two halves (images 0-7, 8-15) of a row's 64 bytes leave back to back (1.81 -> 1.79 ms)
#pragma unroll
        for (int tp = 0; tp < IT / 2; ++tp)
          if (tp * 2 + (kq >> 1) < nPos) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int ti = tp * 2 + h;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (0.0f < acc[ct][ti][e]) ? acc[ct][ti][e] : 0.0f;
                *reinterpret_cast<f32x4*>(dst + ((size_t)tp * 2 * p.Ct + (cc * CT + ct) * 16) * PANEL + 8 * h) = v;
                __builtin_amdgcn_sched_barrier(0);
              }
          }
      } else {
#pragma unroll
        for (int ti = 0; ti < IT; ++ti)
          if ((ti >> 1) * 2 + (kq >> 1) < nPos) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
              *reinterpret_cast<f32x4*>(dst + ((size_t)(ti >> 1) * 2 * p.Ct + (cc * CT + ct) * 16) * PANEL + 8 * (ti & 1)) = acc[ct][ti];
          }
      }
    }
   };
    if (edge) body(std::true_type{}); else body(std::false_type{});
  }
}

// rows: [kh][kw][1][rowStride] slot bytes; ctrd: [Cs][K]; out: [step][S / 16][4 kq][16]: the code word of window element
// k = 4 step + kq = (c knl + kh) knl + kw, zero past the window
__global__ void k_decode_weights_nchw(const uint8_t* __restrict__ rows, const float* __restrict__ ctrd, float* __restrict__ out,
                                      QkSlots sl, int knl, int Cin, int K, int Ct, int steps, int S) {
  const int total = steps * 4 * S;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ch = (i % 16) + 16 * ((i / 64) % (S / 16)), kq = (i / 16) % 4, step = i / (4 * S);
    const int k = 4 * step + kq;
    float w = 0.0f;
    if (ch < Ct && k < Cin * knl * knl) {
      const int kw = k % knl, kh = (k / knl) % knl, c = k / (knl * knl);
      const int slot = rows[(size_t)(kh * knl + kw) * sl.rowStride + qk_slot_entry(sl, 0, ch)];
      w = ctrd[(size_t)c * K + qcnn_row_slot(slot)];
    }
    out[i] = w;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// FC layers whose sub-spaces have ONE dim (a 1000-way classifier behind 4096 features: AlexNet / VGG-16 fc8, 16 code
// words of one float each): a look-up there stands for one multiply-add, so the table build — 4096 sub-spaces x 16 code
// words x 128 images per panel — is pure overhead.  out[c] = bias[c] + sum_k x[k] * w[k][c], w[k][c] = ctrd[k][asmt[k][c]].
//
// A workgroup = 64 channels x 64 images of one panel x one slice of the k axis; its 16 waves take 16 sub-slices, each
// with a 4 x 4 block of accumulator tiles fed by ONE dwordx4 of code words (channel tile t = the channels 4 i + t) and
// ONE dwordx4 of activations (image tile t = the images 4 i + t) per four k; the 16 partial blocks are added through LDS
// in a fixed tree (w + (w + 8), then + 4, + 2, + 1).  gridDim.z > 1: k slices over workgroups too, partial sums to
// scratch for k_sum_partials (few panels: one GPU's share of a sharded batch).
__global__ __launch_bounds__(1024) void k_fc_dec(FcDecParams p) {
  extern __shared__ __attribute__((aligned(16))) float ldsR[];          // 8 waves x 16 tiles x 256 floats
  const int lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
  const int li = lane & 15, kq = lane >> 4;
  const int cb = blockIdx.x, half = blockIdx.y % p.halves, panel = blockIdx.y / p.halves, z = blockIdx.z;
  const int steps = p.D / (4 * 16 * (int)gridDim.z);                    // steps of four k per wave
  const int k0 = (z * 16 + wave) * steps * 4;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wdec), 0, (unsigned)((size_t)p.D * p.S * sizeof(float)), 0x00020000);
  const float* __restrict__ xb = p.src + (size_t)panel * p.D * PANEL;
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, (unsigned)((size_t)p.D * PANEL * sizeof(float)), 0x00020000);
  const int aLane = (kq * p.S + cb * 64 + 4 * li) * (int)sizeof(float);
  const int bLane = (kq * PANEL + half * 64 + 4 * li) * (int)sizeof(float);
  const int aStep = 4 * p.S * (int)sizeof(float), bStep = 4 * PANEL * (int)sizeof(float);
  f32x4 acc[4][4];                                                      // [channel tile][image tile]
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) acc[ct][ti] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  constexpr int R = 3;
  f32x4 a[R], b[R];
  int aoff = k0 * p.S * (int)sizeof(float), boff = k0 * PANEL * (int)sizeof(float);
  auto next = [&](f32x4& aa, f32x4& bb) {                               // past the slice: rows of the next slice, never multiplied
    aa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, aLane, aoff, 0));
    bb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, bLane, boff, 0));
    aoff += aStep; boff += bStep;
  };
  auto products = [&](const f32x4& aa, const f32x4& bb) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) acc[ct][ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ct], bb[ti], acc[ct][ti], 0, 0, 0);
  };
#pragma unroll
  for (int u = 0; u < R; ++u) next(a[u], b[u]);
  int t = 0;
  for (; t + R <= steps; t += R) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      products(a[u], b[u]);
      __builtin_amdgcn_sched_barrier(0);
      next(a[u], b[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < R - 1; ++u)
    if (t + u < steps) products(a[u], b[u]);
  // fixed-order tree over the 16 waves
  f32x4* lds4 = reinterpret_cast<f32x4*>(ldsR);
#pragma unroll
  for (int stride = 8; stride >= 1; stride >>= 1) {
    if (wave >= stride && wave < 2 * stride) {
#pragma unroll
      for (int i = 0; i < 16; ++i) lds4[((wave - stride) * 16 + i) * 64 + lane] = acc[i >> 2][i & 3];
    }
    __syncthreads();
    if (wave < stride) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x4 v = lds4[(wave * 16 + i) * 64 + lane];
        acc[i >> 2][i & 3] += v;
      }
    }
    __syncthreads();
  }
  if (wave != 0) return;
  // tile ct, row 4 kq + r: channel 64 cb + 4 (4 kq + r) + ct; tile ti, column li: image 64 half + 4 li + ti
  float* __restrict__ out = (gridDim.z > 1 ? p.partial + (size_t)z * p.panels * p.Ct * PANEL : p.dst) +
                            (size_t)panel * p.Ct * PANEL + half * 64 + 4 * li;
  const bool first = z == 0;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = cb * 64 + 4 * (4 * kq + r) + ct;
      if (ch < p.Ct) {
        const float bv = first ? p.bias[ch] : 0.0f;
        f32x4 v = {acc[ct][0][r] + bv, acc[ct][1][r] + bv, acc[ct][2][r] + bv, acc[ct][3][r] + bv};
        if (p.relu && gridDim.z == 1) {
          v[0] = (0.0f < v[0]) ? v[0] : 0.0f; v[1] = (0.0f < v[1]) ? v[1] : 0.0f;
          v[2] = (0.0f < v[2]) ? v[2] : 0.0f; v[3] = (0.0f < v[3]) ? v[3] : 0.0f;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)ch * PANEL) = v;
      }
    }
}

// rows: [M = D][rowStride] slot bytes (QkSlots order); ctrd: [M][1][K]; out: [D][S]
__global__ void k_decode_fc_weights(const uint8_t* __restrict__ rows, const float* __restrict__ ctrd, float* __restrict__ out,
                                    QkSlots sl, int D, int K, int Ct, int S) {
  const int G = qcnn_stage_group(K);
  const size_t total = (size_t)D * S;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % S), k = (int)(i / S);
    float w = 0.0f;
    if (ch < Ct) {
      const int slot = rows[(size_t)k * sl.rowStride + qk_slot_entry(sl, 0, ch)];
      w = ctrd[(size_t)k * K + (qcnn_row_slot(slot) - (k % G) * K)];    // stage row = (m % G) * K + code word
    }
    out[i] = w;
  }
}

}  // namespace

bool qk_conv_dec_shape(int Cin, int grp, int M, int Ct, int knl, int* Kp, int* S) {
  if (grp != 1 || M != 1 || Cin < 1 || Cin > 4) return false;
  if (Ct < 32 || Ct % 32) return false;                   // a wave owns 32, 64 or 96 channels
  const int kp = (knl * Cin + 3) / 4 * 4;
  // channel stride: four consecutive k rows of 16 channels on distinct banks (stride = 16 or 48 mod 64) when all code
  // words still fit the LDS that way, else dense rows (AlexNet conv1: 11 x 36 x 96 floats = 152 KB; its A reads then are
  // two-way bank conflicts, 8 instead of 4 LDS cycles per ds_read2_b32 — a sixth of the LDS pipe)
  const size_t lim = 160 * 1024;
  int s = Ct;
  while ((s & 63) != 16 && (s & 63) != 48) s += 16;
  if ((size_t)knl * kp * s * sizeof(float) > lim) s = Ct;
  if ((size_t)knl * kp * s * sizeof(float) > lim) return false;
  *Kp = kp; *S = s;
  return true;
}

hipError_t qk_decode_weights(const uint8_t* rows, const float* ctrd, float* out, const QkSlots& sl, int knl, int Cin, int K,
                             int Ct, int Kp, int S, hipStream_t st) {
  const int total = knl * Kp * S;
  hipLaunchKernelGGL(k_decode_weights, dim3((total + 255) / 256), dim3(256), 0, st, rows, ctrd, out, sl, knl, Cin, K, Ct, Kp, S);
  return hipGetLastError();
}

hipError_t qk_conv_dec(const DecParams& p, hipStream_t st) {
  if ((long long)p.H * p.W * p.Cin * QCNN_PANEL * 4 >= (1ll << 31)) return hipErrorInvalidValue;   // 32-bit BYTE offsets inside a panel
  if (p.Ct % 32) return hipErrorInvalidValue;
  // channels per wave x positions per wave x operand sets in flight x image tiles: as many accumulator tiles as 128
  // registers hold, all channels in one wave where they fit (a chunk of the channels = the B rows loaded once more)
  const bool clamped = p.pad || p.Kr < 4;     // per-lane window clamps: padded layers, and kernel rows of fewer than four products
  if (p.live <= 16) {               // at most one image tile: 16-image items (four positions of 32 channels for padded layers)
    if (clamped) return launch_dec<2, 4, true, 2, 1>(p, st);
    if (p.Ct % 96 == 0) return launch_dec<6, 2, false, 3, 1>(p, st);
    if (p.Ct % 64 == 0) return launch_dec<4, 4, false, 3, 1>(p, st);
    return launch_dec<2, 6, false, 3, 1>(p, st);
  }
  if (clamped) {
    if (p.Ct % 64 == 0) return launch_dec<4, 1, true, 2, 4>(p, st);
    return launch_dec<2, 1, true, 3, 4>(p, st);
  }
  if (p.Ct % 96 == 0) {
    // a launch of a few thousand 96-channel items (one panel: 6050 on 4096 waves = two rounds for 1.5 rounds of work) runs
    // half-items of 48 channels instead (three rounds of half the length; B rows loaded twice)
    const long long items = (long long)p.panels * p.Ho * p.Wo * ((p.live + 63) / 64) * (p.Ct / 96);
    if (items > 4096 && items < 2 * 4096) return launch_dec<3, 1, false, 3, 4>(p, st);
    return launch_dec<6, 1, false, 2, 4>(p, st);
  }
  if (p.Ct % 64 == 0) return launch_dec<4, 1, false, 3, 4>(p, st);
  return launch_dec<2, 2, false, 3, 4>(p, st);
}

bool qk_fc_dec_shape(int D, int M, int Cs, int Ct, int* S) {
  if (Cs != 1 || M != D || D % 64 || Ct < 1) return false;
  if ((size_t)D * PANEL * sizeof(float) >= (1ull << 32)) return false;
  *S = (Ct + 63) / 64 * 64;
  return (size_t)D * *S * sizeof(float) < (1ull << 32);
}

hipError_t qk_decode_fc_weights(const uint8_t* rows, const float* ctrd, float* out, const QkSlots& sl, int D, int K, int Ct,
                                int S, hipStream_t st) {
  hipLaunchKernelGGL(k_decode_fc_weights, dim3(2048), dim3(256), 0, st, rows, ctrd, out, sl, D, K, Ct, S);
  return hipGetLastError();
}

int qk_fc_dec_slices(int D, int Ct, int panels, int live) {
  // k slices over workgroups: until the launch has about a workgroup per CU, every wave keeping >= 4 steps
  const int wgs = ((Ct + 63) / 64) * panels * ((live + 63) / 64);
  int z = 1;
  while (wgs * z < 192 && D % (64 * 2 * z) == 0 && D / (64 * 2 * z) >= 4 && 2 * z <= 32) z *= 2;   // <= 32 slabs of scratch per sub-batch
  return z;
}

hipError_t qk_fc_dec(const FcDecParams& p, int slices, int live, hipStream_t st) {
  if (p.D % (64 * slices)) return hipErrorInvalidValue;
  FcDecParams q = p;
  q.halves = (live + 63) / 64;
  const dim3 grid((unsigned)((p.Ct + 63) / 64), (unsigned)(p.panels * q.halves), (unsigned)slices);
  const size_t shm = 8 * 16 * 256 * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fc_dec), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_fc_dec, grid, dim3(1024), shm, st, q);
  return hipGetLastError();
}

bool qk_conv_dec_nchw_shape(int Cin, int grp, int M, int Ct, int knl, int pad, int* Kp, int* S) {
  if (grp != 1 || M != 1 || Cin < 1 || Cin > 4 || pad != 0 || Ct % 96) return false;
  const int kp = (Cin * knl * knl + 15) / 16 * 16;            // steps in fours
  if ((size_t)(kp + 4) * Ct * sizeof(float) + (kp / 4 + 4) * 16 > 160 * 1024) return false;   // code words + one step of slack + offset table
  *Kp = kp; *S = Ct;
  return true;
}

hipError_t qk_decode_weights_nchw(const uint8_t* rows, const float* ctrd, float* out, const QkSlots& sl, int knl, int Cin, int K,
                                  int Ct, int Kp, int S, hipStream_t st) {
  const int total = Kp * S;
  hipLaunchKernelGGL(k_decode_weights_nchw, dim3((total + 255) / 256), dim3(256), 0, st, rows, ctrd, out, sl, knl, Cin, K, Ct, Kp / 4, S);
  return hipGetLastError();
}

// p.Kr = Cin knl^2, p.Kp = qk_conv_dec_nchw_shape's, p.S = Ct
hipError_t qk_conv_dec_nchw(const DecParams& p, hipStream_t st) {
  if (!p.srcNchw || p.pad != 0 || p.Ct % 96 || p.S != p.Ct || (unsigned long long)p.nImages * p.Cin * p.H * p.W * 4ull >= (1ull << 32))
    return hipErrorInvalidValue;
  const long long items = (long long)p.panels * p.Ho * ((p.Wo + 3) / 4) * ((p.live + 15) / 16) * (p.Ct / 96);
  const int blocks = (int)std::min<long long>(256, items);           // (few items: one per workgroup, see `sparse`)
  const size_t shm = (size_t)(p.Kp + 4) * p.S * sizeof(float) + (size_t)(p.Kp / 4 + 4) * 16;
  auto kern = k_conv_dec_nchw<6, 4>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NCHW_WAVES), shm, st, p);
  return hipGetLastError();
}
