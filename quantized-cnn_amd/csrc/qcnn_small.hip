// qcnn_small.hip — the approximate conv / FC layers for batches of a few images (the reference's own regime is
// ONE image per forward pass: src/CaffeEva.cc:23, ExecForwardPass(img, prob) :213-261).
//
// The panel kernels (qcnn_kernels.hip) put 128 images on the lanes of a wave: a single image pays for a whole
// panel (2.8 ms).  Here the lanes are OUTPUT CHANNELS and positions instead, one workgroup per (output tile,
// channel chunk, image):
//   per chunk of sub-spaces:  stage the receptive field's activations and the chunk's assignment bytes in LDS, build
//                             the tile's look-up table  LUT[source pixel][sub-space][code word]  in LDS on the matrix pipe
//                             (GetInPdMat, src/CaffeEva.cc:1261-1296, for the pixels of the tile's receptive field),
//                             then every thread walks the taps of its outputs and gathers
//                             acc += LUT[pixel(tap)][m][assignment]   (CalcFeatMap_ConvAprx :840-863 / _FCntAprx :998-1023).
// Activations stay in the panel layout ([E][128], image = lane index) so that the glue kernels and the large-batch
// path interoperate; the assignment tables are the same one-byte row slots the panel kernels use (the code-word index
// is recovered from the slot): one byte per look-up is streamed, as in the reference.  Summation runs over sub-space chunks first, so results agree
// with the panel kernels / the reference to rounding (~1e-6), not bit for bit; the exact builder (QCNN_OPT_LUT_MODE = 0)
// therefore always takes the panel kernels.
#include "qcnn_kernels.h"

#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PANEL = QCNN_PANEL;
constexpr int NT = 512;                     // threads per workgroup
constexpr int LUT_BYTES = 112 * 1024;       // LDS given to the FC table chunk (k_fc_small)
constexpr int CONV_LDS = 156 * 1024;        // conv: table chunk + staged activations + staged assignments of a sub-space chunk

// inverse of qcnn_row_slot (the permutation swaps two 2-bit fields: it is its own inverse)
__device__ __forceinline__ int slot_row(int s) { return (s & 0x70) | ((s & 3) << 2) | ((s >> 2) & 3); }

struct SmallConv {
  const float* src;        // panels [H*W*Cin][128]  (srcNchw = 0)  or the NCHW network input (srcNchw = 1)
  float* dst;              // panels [Ho*Wo*Ct][128]
  const float* bias;
  const float* ctrd;       // [M][Cs][K]
  const uint8_t* rows;     // [taps][M][rowStride]: row slots
  int srcNchw, img0;       // img0: index of image 0 of this launch inside the batch (NCHW addressing)
  int H, W, Cin, Ho, Wo, Ct, knl, stride, pad, grp;
  int M, Cs, K, G, relu, rowStride;
  int TH, TW, tilesX, CH, chunks, MC;   // output tile, channels per workgroup, chunks per group, sub-spaces per LUT chunk
  int lutFloats, xsFloats;              // LDS: table [npx][MC][K], activations [npx][MC * Cs], then assignments [taps][MC][CH] bytes
  QkSlots sl;
};

__device__ __forceinline__ float load_x(const SmallConv& p, int img, int hi, int wi, int ch) {
  if (p.srcNchw) return p.src[(((size_t)(p.img0 + img) * p.Cin + ch) * p.H + hi) * p.W + wi];
  const int panel = img / PANEL, lane = img % PANEL;
  return p.src[((size_t)panel * p.H * p.W * p.Cin + (size_t)(hi * p.W + wi) * p.Cin + ch) * PANEL + lane];
}

// one look-up: row slot -> stage row -> code word of sub-space m -> table entry
__device__ __forceinline__ float lut_at(const float* __restrict__ tab, uint8_t slot, int mInStage, int K) {
  return tab[slot_row(slot) - mInStage * K];
}

__global__ __launch_bounds__(NT) void k_conv_small(SmallConv p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ lut = lds;                                   // [npx][MC][K]
  float* __restrict__ xs = lds + p.lutFloats;                      // [npx][MC * Cs] activations of the current chunk
  uint8_t* __restrict__ idx = reinterpret_cast<uint8_t*>(xs + p.xsFloats);   // [taps][mc][CH] row slots of the current chunk
  const int t = threadIdx.x;
  const int img = blockIdx.z;
  const int ty = blockIdx.x / p.tilesX, tx = blockIdx.x % p.tilesX;
  const int grp = blockIdx.y / p.chunks, chunk = blockIdx.y % p.chunks;
  const int Cg = p.Cin / p.grp, Ctg = p.Ct / p.grp;
  const int ho0 = ty * p.TH, wo0 = tx * p.TW;
  const int hoL = min(ho0 + p.TH, p.Ho) - 1, woL = min(wo0 + p.TW, p.Wo) - 1;
  const int hiL = max(0, ho0 * p.stride - p.pad), hiU = min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1);
  const int wiL = max(0, wo0 * p.stride - p.pad), wiU = min(p.W - 1, woL * p.stride - p.pad + p.knl - 1);
  const int rfW = wiU - wiL + 1, npx = (hiU - hiL + 1) * rfW;
  const int K = p.K, Cs = p.Cs;

  // thread -> one output channel and up to 4 positions of the tile
  const int slots = NT / p.CH;                          // position slots
  const int cl = t % p.CH, pslot = t / p.CH;
  const int cg = chunk * p.CH + cl;                     // channel inside the group
  const bool chOk = pslot < slots && cg < Ctg;
  const int c = grp * Ctg + cg;
  const int NP = p.TH * p.TW;
  float acc[4];
  const float b = chOk ? p.bias[c] : 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = b;

  for (int m0 = 0; m0 < p.M; m0 += p.MC) {
    const int mc = min(p.MC, p.M - m0);
    const int dims = mc * Cs;                            // activation dims of this chunk (those past Cg are zero)
    // ---- stage the receptive field's activations of this chunk (independent loads, a few per thread)
    for (int e = t; e < npx * dims; e += NT) {
      const int px = e / dims, d = e % dims;
      const int ch = m0 * Cs + d;
      xs[e] = (ch < Cg) ? load_x(p, img, hiL + px / rfW, wiL + px % rfW, grp * Cg + ch) : 0.0f;
    }
    // ---- stage the chunk's assignments of this workgroup's channels: the gather below then never waits for HBM / L2 (a
    //      look-up used to be a one-byte global load followed by the table read that depends on it: the whole kernel ran
    //      at the latency of that chain)
    {
      // NT is a multiple of CH: a thread always stages the bytes of ONE channel (its table position is computed once);
      // sixteen independent loads in flight per thread, then the stores
      const int c2 = t % p.CH, prStep = NT / p.CH, npair = p.knl * p.knl * mc;
      const int cg2 = chunk * p.CH + c2;
      const bool ok2 = cg2 < Ctg;
      const uint8_t* __restrict__ rbase = p.rows + (size_t)m0 * p.rowStride + (ok2 ? qk_slot_entry(p.sl, grp, cg2) : 0);
      for (int pair = t / p.CH; pair < npair; pair += prStep * 16) {
        uint8_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int pp = min(pair + u * prStep, npair - 1);
          v[u] = rbase[((size_t)(pp / mc) * p.M + pp % mc) * p.rowStride];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int pp = pair + u * prStep;
          if (pp < npair) idx[pp * p.CH + c2] = ok2 ? v[u] : (uint8_t)0;
        }
      }
    }
    __syncthreads();
    // ---- build: a thread owns a (sub-space, code word) pair, keeps the code word in registers and walks pixels; when
    //      there are fewer pairs than threads (first layer: one sub-space) the pixels are dealt out to NT / pairs thread
    //      groups.  Activation reads are LDS broadcasts, table writes are consecutive in k.
    if (K % 16 == 0) {
      // Matrix build: the chunk's table is a small product [pixels x dims] x [dims x K] per sub-space — 16 x 16 tiles of
      // v_mfma_f32_16x16x4_f32 dealt out to the eight waves (A = staged activations out of LDS, B = the code book rows,
      // 64-byte segments from L2).  The scalar build below took a third of the kernel's time for one image.
      const int wave = t >> 6, lane = t & 63, li = lane & 15, kq = lane >> 4;
      const int rowTiles = (npx + 15) >> 4, colTiles = K >> 4;
      for (int tile = wave; tile < mc * rowTiles * colTiles; tile += NT / 64) {
        const int mloc = tile / (rowTiles * colTiles), rt = (tile / colTiles) % rowTiles, ctile = tile % colTiles;
        const int m = m0 + mloc;
        const int dsel = min(Cg - m * Cs, Cs);
        const int pxA = rt * 16 + li;
        f32x4 accT = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int d0 = 0; d0 < dsel; d0 += 4) {
          const int d = d0 + kq;
          const float av = (pxA < npx && d < dsel) ? xs[pxA * dims + mloc * Cs + d] : 0.0f;
          const float bv = (d < dsel) ? p.ctrd[((size_t)m * Cs + d) * K + ctile * 16 + li] : 0.0f;
          accT = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accT, 0, 0, 0);
        }
        const int kc = ctile * 16 + li;
        const int kst = (p.G == 1) ? qcnn_row_slot(kc) : kc;   // K = 128: entries in ROW-SLOT order, a look-up is tab[byte]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = rt * 16 + 4 * kq + r;                  // D[4 kq + r][li]
          if (px < npx) lut[(px * p.MC + mloc) * K + kst] = accT[r];
        }
      }
    } else {
      const int npairs = mc * K;
      const int groups = npairs < NT ? NT / npairs : 1;
      const int pg = npairs < NT ? t / npairs : 0;
      for (int pair = (npairs < NT ? t % npairs : t); pair < npairs && pg < groups; pair += NT) {
        const int k = pair % K, mloc = pair / K;
        const int kst = (p.G == 1) ? qcnn_row_slot(k) : k;   // K = 128: entries in ROW-SLOT order, a look-up is tab[byte]
        const int m = m0 + mloc;
        const int dsel = min(Cg - m * Cs, Cs);
        float cw[QCNN_MAX_CS];
#pragma unroll
        for (int d = 0; d < QCNN_MAX_CS; ++d) cw[d] = (d < dsel) ? p.ctrd[((size_t)m * Cs + d) * K + k] : 0.0f;
        const float* __restrict__ xr = xs + mloc * Cs;
        for (int px = pg; px < npx; px += groups) {
          float v = 0.0f;
#pragma unroll
          for (int d = 0; d < QCNN_MAX_CS; ++d)
            if (d < dsel) v = fmaf(xr[px * dims + d], cw[d], v);
          lut[(px * p.MC + mloc) * K + kst] = v;
        }
      }
    }
    __syncthreads();
    // ---- gather
    if (chOk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = pslot + j * slots;
        if (q >= NP) break;
        const int ho = ho0 + q / p.TW, wo = wo0 + q % p.TW;
        if (ho >= p.Ho || wo >= p.Wo) continue;
        const int hs = ho * p.stride - p.pad, ws = wo * p.stride - p.pad;
        const int khL = max(0, -hs), khU = min(p.knl - 1, p.H - 1 - hs);
        const int kwL = max(0, -ws), kwU = min(p.knl - 1, p.W - 1 - ws);
        float a = acc[j];
        for (int kh = khL; kh <= khU; ++kh) {
          const float* rowTab = lut + (ptrdiff_t)((hs + kh - hiL) * rfW + (ws - wiL)) * (p.MC * K);
          const uint8_t* rowIdx = idx + (size_t)(kh * p.knl) * mc * p.CH + cl;      // [kw][ml][CH]
          // batches of eight INDEPENDENT look-ups (offset loads in flight together, then the table reads, then the adds in
          // order); the tail of a batch re-reads the last valid element and is not added
          if (mc == 1) {                                 // one sub-space per pixel (first layer): run over the taps
            const int mi = m0 % p.G;
            for (int kw = kwL; kw <= kwU; kw += 8) {
              uint8_t o[8];
              float v[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) o[u] = rowIdx[min(kw + u, kwU) * mc * p.CH];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const float* tb = rowTab + (ptrdiff_t)min(kw + u, kwU) * (p.MC * K);
                v[u] = (p.G == 1) ? tb[o[u]] : lut_at(tb, o[u], mi, K);
              }
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (kw + u <= kwU) a += v[u];
            }
          } else {
            for (int kw = kwL; kw <= kwU; ++kw) {
              const float* tab = rowTab + (ptrdiff_t)kw * (p.MC * K);
              const uint8_t* rr = rowIdx + kw * mc * p.CH;
              for (int ml = 0; ml < mc; ml += 8) {
                uint8_t o[8];
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = rr[min(ml + u, mc - 1) * p.CH];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const int mm = min(ml + u, mc - 1);
                  v[u] = (p.G == 1) ? tab[mm * K + o[u]] : lut_at(tab + mm * K, o[u], (m0 + mm) & (p.G - 1), K);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                  if (ml + u < mc) a += v[u];
              }
            }
          }
        }
        acc[j] = a;
      }
    }
    __syncthreads();
  }
  if (chOk) {
    const int panel = img / PANEL, lane = img % PANEL;
    float* dst = p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = pslot + j * slots;
      if (q >= NP) break;
      const int ho = ho0 + q / p.TW, wo = wo0 + q % p.TW;
      if (ho >= p.Ho || wo >= p.Wo) continue;
      float v = acc[j];
      if (p.relu) v = (0.0f < v) ? v : 0.0f;
      dst[((size_t)(ho * p.Wo + wo) * p.Ct + c) * PANEL] = v;
    }
  }
}

struct SmallFc {
  const float* src;        // panels [D][128] (consumption order)
  float* dst;              // panels [Ct][128]
  float* lut;              // scratch [n][M][K]: the look-up table of every image, built once by k_fc_lut
  const float* bias;
  const float* ctrd;
  const uint8_t* rows;     // [M][rowStride]: row slots
  const uint8_t* cbn;      // or: the .cbn payload ([Ct][M] code words, `bits` each, 4096-byte blocks of `per` values), read in place
  int bits, per;
  int D, Ct, M, Cs, K, G, relu, rowStride, MC;
  QkSlots sl;
};

// GetInPdMat for an FC layer and a few images (src/CaffeEva.cc:1261-1296 with P = 1): one thread per table entry.
// At this batch size the table is tiny (AlexNet fc6: 295 KB per image) — it is materialised once, as the reference
// does, instead of being rebuilt by every workgroup of the gather kernel.
__global__ __launch_bounds__(256) void k_fc_lut(SmallFc p) {
  const int img = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= p.M * p.K) return;
  const int k = e % p.K, m = e / p.K;
  const float* __restrict__ x = p.src + (size_t)(img / PANEL) * p.D * PANEL + (img % PANEL);
  const float* __restrict__ cm = p.ctrd + (size_t)m * p.Cs * p.K + k;
  const int dsel = min(p.D - m * p.Cs, p.Cs);
  float v = 0.0f;
#pragma unroll
  for (int d = 0; d < QCNN_MAX_CS; ++d)
    if (d < dsel) v = fmaf(x[(size_t)(m * p.Cs + d) * PANEL], cm[d * p.K], v);
  p.lut[(size_t)img * p.M * p.K + e] = v;
}

// workgroup = 16 output channels x 32 slices of the sub-space chunk; one image.  The table chunk is staged from the
// materialised table into LDS (coalesced float4 copies), then gathered.
constexpr int FC_CH = 16, FC_SLICES = NT / FC_CH;
__global__ __launch_bounds__(NT) void k_fc_small(SmallFc p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ lut = lds;                                   // [MC][K]
  const int t = threadIdx.x;
  const int img = blockIdx.z;
  const int panel = img / PANEL, lane = img % PANEL;
  const int K = p.K;
  const float* __restrict__ tab = p.lut + (size_t)img * p.M * K;
  const int cl = t % FC_CH, slice = t / FC_CH;
  const int c = blockIdx.x * FC_CH + cl;
  const bool chOk = c < p.Ct;
  const int entry = chOk ? qk_slot_entry(p.sl, 0, c) : 0;
  float acc = 0.0f;
  for (int m0 = 0; m0 < p.M; m0 += p.MC) {
    const int mc = min(p.MC, p.M - m0);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(tab + (size_t)m0 * K);
    float4* l4 = reinterpret_cast<float4*>(lut);
    for (int e = t; e < mc * K / 4; e += NT) l4[e] = g4[e];          // K is a multiple of 4 (checked by the launcher)
    __syncthreads();
    if (chOk) {
      const int per = (mc + FC_SLICES - 1) / FC_SLICES;
      const int a0 = slice * per, a1 = min(mc, a0 + per);
      if (p.cbn != nullptr) {
        // Packed stream in place: the channel's assignments of this slice are CONSECUTIVE values of the
        // file order [Ct][M] — (a1 - a0) x bits contiguous bits, at most one 4096-byte block boundary inside.  A value
        // = the `bits` bits at bit (index in block) x bits, MSB first: two byte loads and a shift (values never
        // straddle a block; a second byte that is not needed is not read past the block).
        if (a0 < a1) {
          const size_t e0 = (size_t)c * p.M + (size_t)(m0 + a0);
          const uint8_t* __restrict__ blk = p.cbn + (e0 / (size_t)p.per) * 4096;
          const int r0 = (int)(e0 % (size_t)p.per);
          const unsigned mask = (1u << p.bits) - 1u;
          for (int ml = a0; ml < a1; ml += 8) {                       // eight independent look-ups at a time
            unsigned code[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              int rr = r0 + (min(ml + u, a1 - 1) - a0);
              const uint8_t* __restrict__ b = blk;
              if (rr >= p.per) { rr -= p.per; b += 4096; }
              const int bit0 = rr * p.bits;
              const uint8_t* __restrict__ q = b + (bit0 >> 3);
              const unsigned w = ((unsigned)q[0] << 8) | (unsigned)q[(bit0 & 7) + p.bits > 8 ? 1 : 0];
              code[u] = (w >> (16 - (bit0 & 7) - p.bits)) & mask;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = lut[min(ml + u, a1 - 1) * K + (int)code[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (ml + u < a1) acc += v[u];
          }
        }
      } else {
      const uint8_t* rr = p.rows + (size_t)m0 * p.rowStride + entry;
      for (int ml = a0; ml < a1; ml += 8) {                         // eight independent look-ups at a time
        uint8_t o[8];
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = rr[(size_t)min(ml + u, a1 - 1) * p.rowStride];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int mm = min(ml + u, a1 - 1);
          v[u] = lut_at(lut + mm * K, o[u], (m0 + mm) % p.G, K);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (ml + u < a1) acc += v[u];
      }
      }
    }
    __syncthreads();
  }
  float* red = lds;                                      // the table is dead: reuse its first 2 KB
  red[t] = acc;
  __syncthreads();
  if (slice == 0 && chOk) {
    float v = p.bias[c];
    for (int s2 = 0; s2 < FC_SLICES; ++s2) v += red[s2 * FC_CH + cl];
    if (p.relu) v = (0.0f < v) ? v : 0.0f;
    p.dst[((size_t)panel * p.Ct + c) * PANEL + lane] = v;
  }
}

hipError_t allow_lds(const void* kern, int bytes) {
  return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

// Tile choice: channels per workgroup CH = up to 128 of the group's channels (512 threads = CH x 4 position slots,
// 4 positions per slot), the largest square-ish output tile whose receptive field x sub-space chunk fits the LDS table
// with at least min(M, 4) sub-spaces per chunk — small tiles are preferred anyway: one image has to spread over 256 CUs.
hipError_t qk_conv_small(const ConvParams& cp, int n, hipStream_t st) {
  SmallConv p;
  p.src = cp.src; p.dst = cp.dst; p.bias = cp.bias; p.ctrd = cp.ctrd; p.rows = cp.rows;
  p.srcNchw = cp.srcNchw; p.img0 = cp.panel0 * PANEL;
  p.H = cp.H; p.W = cp.W; p.Cin = cp.Cin; p.Ho = cp.Ho; p.Wo = cp.Wo; p.Ct = cp.Ct;
  p.knl = cp.knl; p.stride = cp.stride; p.pad = cp.pad; p.grp = cp.grp;
  p.M = cp.M; p.Cs = cp.Cs; p.K = cp.K; p.G = qcnn_stage_group(cp.K); p.relu = cp.relu;
  const int Ctg = cp.Ct / cp.grp;
  p.sl = qk_conv_slots(Ctg, cp.grp);
  p.rowStride = p.sl.rowStride;
  p.CH = std::min(128, (Ctg + 31) / 32 * 32);
  p.chunks = (Ctg + p.CH - 1) / p.CH;
  const int slots = NT / p.CH;
  // output tile: 2 x 2 unless the map is tiny; shrink until the table chunk holds >= min(M, 4) sub-spaces
  int th = std::min(2, cp.Ho), tw = std::min(2, cp.Wo);
  auto rf = [&](int a) { return (a - 1) * cp.stride + cp.knl; };
  // sub-spaces per chunk: the table [npx][mc][K], the staged activations [npx][mc * Cs] and the staged assignments
  // [taps][mc][CH] have to fit
  auto mcFor = [&](int a, int b) {
    return CONV_LDS / (rf(a) * rf(b) * (cp.K + cp.Cs) * 4 + cp.knl * cp.knl * p.CH);
  };
  while ((mcFor(th, tw) < std::min(cp.M, 4) || th * tw > 4 * slots) && (th > 1 || tw > 1)) {
    if (tw >= th && tw > 1) --tw; else --th;
  }
  if (mcFor(th, tw) < 1) return hipErrorInvalidValue;      // a single pixel's window does not fit: not a small-path layer
  p.TH = th; p.TW = tw;
  p.MC = std::min(cp.M, mcFor(th, tw));
  p.lutFloats = rf(th) * rf(tw) * p.MC * cp.K;
  p.xsFloats = rf(th) * rf(tw) * p.MC * cp.Cs;
  const int ldsBytes = (p.lutFloats + p.xsFloats) * 4 + cp.knl * cp.knl * p.MC * p.CH;
  p.tilesX = (cp.Wo + tw - 1) / tw;
  const int tilesY = (cp.Ho + th - 1) / th;
  hipError_t e = allow_lds(reinterpret_cast<const void*>(k_conv_small), ldsBytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_conv_small, dim3(p.tilesX * tilesY, p.chunks * cp.grp, n), dim3(NT), ldsBytes, st, p);
  return hipGetLastError();
}

// scratch: n * M * K floats (the materialised tables); the engine passes its FC partial-sum buffer
hipError_t qk_fc_small(const FcParams& fp, int n, hipStream_t st) {
  if (fp.K % 4 || fp.partial == nullptr) return hipErrorInvalidValue;
  SmallFc p;
  p.src = fp.src; p.dst = fp.dst; p.lut = fp.partial; p.bias = fp.bias; p.ctrd = fp.ctrd; p.rows = fp.rows;
  p.cbn = (fp.cbnBits >= 1 && fp.cbnBits <= 8) ? fp.cbn : nullptr;
  p.bits = fp.cbnBits; p.per = p.cbn ? 4096 * 8 / fp.cbnBits : 1;
  p.D = fp.D; p.Ct = fp.Ct; p.M = fp.M; p.Cs = fp.Cs; p.K = fp.K; p.G = qcnn_stage_group(fp.K); p.relu = fp.relu;
  p.sl = qk_fc_slots(fp.Ct);
  p.rowStride = p.sl.rowStride;
  p.MC = std::min(fp.M, LUT_BYTES / (fp.K * 4));
  hipLaunchKernelGGL(k_fc_lut, dim3((fp.M * fp.K + 255) / 256, n), dim3(256), 0, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = allow_lds(reinterpret_cast<const void*>(k_fc_small), LUT_BYTES);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_fc_small, dim3((fp.Ct + FC_CH - 1) / FC_CH, 1, n), dim3(NT), LUT_BYTES, st, p);
  return hipGetLastError();
}
