// qcnn_kernels.hip — hand-written gfx950 (CDNA4) kernels of the Quantized-CNN approximate forward pass.
//
// Hot kernels:
//   k_conv_aprx  fused  GetInPdMat (src/CaffeEva.cc:1261-1296)  +  CalcFeatMap_ConvAprx (:760-868)
//   k_fc_aprx    fused  GetInPdMat                              +  CalcFeatMap_FCntAprx (:968-1025)
// Glue kernels (row a9): ReLU :1027, LRN :1038, max-pool :870, softmax :1098, top-5 :1162,
// NCHW<->panel conversions (:1146-1160, :187-189).
//
// Mapping (see qcnn_kernels.h for the HBM layout): a lane carries an image pair.  A workgroup (8 waves)
// owns one 128-image panel, one tile of output positions and one slice of output channels; every wave
// keeps (positions x channels-per-wave) float2 accumulators in VGPRs.  The look-up table is never
// materialised in HBM: it is produced one STAGE at a time in LDS — a stage = 128 code-word rows =
// G = 128/K consecutive sub-spaces of one source pixel (conv) or of the input vector (FC) for the 128
// images; a row is 512 contiguous bytes (+16 B pad) = one conflict-free ds_read_b64 per wave.  Stages
// are double buffered: while the waves gather from stage s, the MFMA operands of stage s+1 are already
// in flight and its tiles (v_mfma_f32_16x16x4_f32, or ordered VALU mul+add in "exact" mode) are written
// to the other buffer; one s_barrier per stage.  Stages are visited in (pixel row-major, sub-space
// ascending) order, which for any one output is exactly the reference's (kh, kw, m) summation order
// (:840-863), so with the exact builder conv/FC outputs are bit-identical to the reference.  Code-word
// offsets are wave-uniform and arrive through scalar loads.
#include "qcnn_kernels.h"

#include <float.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PANEL = QCNN_PANEL;              // images per panel
constexpr int NW = 8;                          // waves per workgroup of the two hot kernels
constexpr int ROWB = QCNN_ROW_BYTES;           // LDS bytes per code-word row
constexpr int STAGE_ROWS = QCNN_STAGE_ROWS;
constexpr int STAGE_BYTES = STAGE_ROWS * ROWB;  // 67 584 B; two stages = 132 KB of the 160 KB LDS
constexpr int XROWB = PANEL * 4;               // bytes of one activation row in HBM

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// CPW look-ups of one (position, sub-space): the code-word offsets are wave-uniform and arrive four at a
// time through s_load_dwordx4 (the tables are 16-byte aligned: Ct and the wave's first channel are
// multiples of 4); each look-up is one ds_read_b64 of the image pair + one v_pk_add_f32.
template <int CPW>
__device__ __forceinline__ void gather_row(f32x2 (&acc)[CPW], const uint32_t* __restrict__ ap, const char* stage) {
  static_assert(CPW % 4 == 0, "offsets are fetched as 4 x uint32");
  constexpr int G = (CPW % 24 == 0) ? 24 : ((CPW % 16 == 0) ? 16 : ((CPW % 12 == 0) ? 12 : ((CPW % 8 == 0) ? 8 : 4)));
  const u32x4* __restrict__ ap4 = reinterpret_cast<const u32x4*>(__builtin_assume_aligned(ap, 16));
#pragma unroll
  for (int g0 = 0; g0 < CPW; g0 += G) {
    // phase 1: all scalar loads of the group (SMEM returns out of order, so LDS reads may only be
    // counted with s_waitcnt lgkmcnt(N > 0) once no scalar load is in flight)
    u32x4 o[G / 4];
#pragma unroll
    for (int j = 0; j < G / 4; ++j) o[j] = ap4[g0 / 4 + j];
    __builtin_amdgcn_sched_barrier(0);
    // phase 2: G back-to-back ds_read_b64, then the packed adds as the reads return
    f32x2 v[G];
#pragma unroll
    for (int j = 0; j < G / 4; ++j) {
      v[4 * j + 0] = *reinterpret_cast<const f32x2*>(stage + o[j].x);
      v[4 * j + 1] = *reinterpret_cast<const f32x2*>(stage + o[j].y);
      v[4 * j + 2] = *reinterpret_cast<const f32x2*>(stage + o[j].z);
      v[4 * j + 3] = *reinterpret_cast<const f32x2*>(stage + o[j].w);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) acc[g0 + j] += v[j];
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------------------
// LUT stage builders.  A stage covers sub-spaces m0 .. m0+G-1 (those < mEnd), K rows each.  The
// 128-image activation row of dim d of sub-space m starts at xbase + xoff0 + (m*Cs + d) * 512 bytes.
// ------------------------------------------------------------------------------------------------

// exact: y = ((0 + x0*c0) + x1*c1) + ...  with separately rounded product and sum, the order of the
// reference's saxpy chain (src/CaffeEva.cc:1284-1289, include/BlasWrapper.h:164-184).  Any K <= 128.
__device__ __forceinline__ void build_stage_exact(char* stage, const char* __restrict__ xbase, uint32_t xoff0,
                                                  const float* __restrict__ ctrd, int K, int Cs, int D, int G, int m0,
                                                  int mEnd, int wave, int lane) {
  const int kpw = (K + NW - 1) / NW;
  const int k0 = wave * kpw;
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
// tiles; wave w owns image tile w and all 8 row tiles.  Row tile i belongs to sub-space m0 + (16 i)/K
// and starts at code word (16 i) % K  (KT = K/16 in {1, 2, 4, 8}).  Operands are fetched into registers
// (mfma_load) one stage ahead of their use (mfma_store).  All operand loads are UNCONDITIONAL, at
// wave-uniform base + per-lane constant + immediate (device buffers carry slack for the over-read of
// dims / sub-spaces that do not exist); what must not contribute is zeroed by a select.
template <int KT>
struct MfmaOps {
  static constexpr int NB = (KT == 8) ? 1 : 8;   // K = 128: one sub-space per stage, one activation operand
  float a[8][2];    // code-book operand per row tile and k-step
  float b[NB][2];   // activation operand per row tile (sub-space) and k-step
};

template <int KT>
__device__ __forceinline__ void mfma_load(MfmaOps<KT>& o, const char* __restrict__ xbase, uint32_t xoff0,
                                          const float* __restrict__ ctrd, int Cs, int D, int m0, int mEnd, int wave,
                                          int lane) {
  constexpr int K = KT * 16;
  constexpr int NB = MfmaOps<KT>::NB;
  constexpr int SUBS = 128 / K;                               // sub-spaces per stage
  const int li = lane & 15, lk = lane >> 4;
  const float* __restrict__ cb = ctrd + (size_t)m0 * Cs * K + (lk * K + li);                          // + uniform
  const char* __restrict__ xb = xbase + xoff0 + (uint32_t)(m0 * Cs) * (uint32_t)XROWB + (lk * XROWB + (wave * 16 + li) * 4);
  const int ksteps = (min(D, Cs) > 4) ? 2 : 1;
  (void)mEnd; (void)SUBS;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if (ks < ksteps) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int mi = (i * 16) / K, kk = (i * 16) % K;       // compile-time
        o.a[i][ks] = cb[(mi * Cs + ks * 4) * K + kk];
        if (i < NB) o.b[i][ks] = *reinterpret_cast<const float*>(xb + (mi * Cs + ks * 4) * XROWB);
      }
    }
  }
}

// The stage described by (m0, mEnd, D, Cs) is the one `o` was loaded for: operands of dims / sub-spaces
// that do not exist are zeroed here (at use, so that the loads stay in flight during the gather).
template <int KT>
__device__ __forceinline__ void mfma_store(MfmaOps<KT>& o, char* stage, int ksteps, int Cs, int D, int m0, int mEnd,
                                           int wave, int lane) {
  constexpr int K = KT * 16;
  constexpr int NB = MfmaOps<KT>::NB;
  constexpr int SUBS = 128 / K;
  const int li = lane & 15, lk = lane >> 4;
  // plain: every sub-space of the stage exists and has all Cs (4 or 8) dims -> nothing to zero
  const bool plain = (m0 + SUBS <= mEnd) && (D - (m0 + SUBS - 1) * Cs >= Cs) && (Cs == 4 || Cs == 8);
  if (!plain) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int mi = (i * 16) / K;
        const bool ok = (ks < ksteps) && (m0 + mi < mEnd) && (ks * 4 + lk < min(D - (m0 + mi) * Cs, Cs));
        o.a[i][ks] = ok ? o.a[i][ks] : 0.0f;
        if (i < NB) o.b[i][ks] = ok ? o.b[i][ks] : 0.0f;
      }
    }
  }
  char* w0 = stage + (lk * 4) * ROWB + (wave * 16 + li) * 4;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int h = 0; h < 2; ++h) {             // two batches of four independent tiles
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 4 * h + j;
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][0], o.b[NB == 1 ? 0 : i][0], zero, 0, 0, 0);
    }
    if (ksteps > 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = 4 * h + j;
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][1], o.b[NB == 1 ? 0 : i][1], acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* w = w0 + (4 * h + j) * 16 * ROWB;
      *reinterpret_cast<float*>(w) = acc[j][0];
      *reinterpret_cast<float*>(w + ROWB) = acc[j][1];
      *reinterpret_cast<float*>(w + 2 * ROWB) = acc[j][2];
      *reinterpret_cast<float*>(w + 3 * ROWB) = acc[j][3];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv: TH x TW output positions, 8 waves, CPW channels per wave; KT = K/16 for the MFMA builder,
// KT = 0 selects the exact builder (any K <= 128).
// ------------------------------------------------------------------------------------------------
template <int TH, int TW, int CPW, int KT>
__global__ __launch_bounds__(NW * 64) void k_conv_aprx(ConvParams p, int tilesX, int chunksPerGrp, int G) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NT = TH * TW;
  constexpr int KTT = KT > 0 ? KT : 1;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int ty = blockIdx.x / tilesX, tx = blockIdx.x % tilesX;
  const int g = blockIdx.y / chunksPerGrp, chunk = blockIdx.y % chunksPerGrp;
  const int panel = blockIdx.z;
  const int Cg = p.Cin / p.grp, Ctg = p.Ct / p.grp;
  const int cw0 = chunk * (NW * CPW) + wave * CPW;   // first channel of this wave inside the group
  const int ccnt = min(CPW, Ctg - cw0);              // <= 0: the wave only helps building stages
  const int c0 = g * Ctg + cw0;
  const int K = p.K, M = p.M, Cs = p.Cs;
  const int MG = (M + G - 1) / G;                    // stages per source pixel
  const int MCt = M * p.Ct;
  const int ksteps = (min(Cg, Cs) > 4) ? 2 : 1;      // MFMA k-steps (4 dims each) that carry data

  const char* __restrict__ xbase =
      reinterpret_cast<const char*>(p.src + ((size_t)panel * p.H * p.W * p.Cin + (size_t)g * Cg) * PANEL);

  f32x2 acc[NT][CPW];
  {
    const float* __restrict__ bp = p.bias + c0;   // reads past the last channel stay inside the arena
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const float b = bp[c];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][c] = f32x2{b, b};
    }
  }

  const int ho0 = ty * TH, wo0 = tx * TW;
  const int hoL = min(ho0 + TH, p.Ho) - 1, woL = min(wo0 + TW, p.Wo) - 1;   // last real position of the tile
  const int hiL = max(0, ho0 * p.stride - p.pad), hiU = min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1);
  const int wiL = max(0, wo0 * p.stride - p.pad), wiU = min(p.W - 1, woL * p.stride - p.pad + p.knl - 1);
  const int S = (hiU - hiL + 1) * (wiU - wiL + 1) * MG;

  // first source row / column of every position of the tile; positions outside the map get a start that
  // can never match a tap
  int rowStart[TH], colStart[TW];
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) rowStart[dy] = (ho0 + dy < p.Ho) ? (ho0 + dy) * p.stride - p.pad : -(1 << 28);
#pragma unroll
  for (int dx = 0; dx < TW; ++dx) colStart[dx] = (wo0 + dx < p.Wo) ? (wo0 + dx) * p.stride - p.pad : -(1 << 28);

  MfmaOps<KTT> ops;
  int hi = hiL, wi = wiL, mg = 0;
  {
    const uint32_t xoff0 = (uint32_t)(hi * p.W + wi) * p.Cin * (uint32_t)XROWB;
    if (KT > 0) {
      mfma_load<KTT>(ops, xbase, xoff0, p.ctrd, Cs, Cg, 0, M, wave, lane);
      mfma_store<KTT>(ops, lds, ksteps, Cs, Cg, 0, M, wave, lane);
    } else {
      build_stage_exact(lds, xbase, xoff0, p.ctrd, K, Cs, Cg, G, 0, M, wave, lane);
    }
  }
  __syncthreads();

  for (int s = 0; s < S; ++s) {
    int mgn = mg + 1, wn = wi, hn = hi;
    if (mgn == MG) {
      mgn = 0;
      if (++wn > wiU) { wn = wiL; ++hn; }
    }
    const bool more = s + 1 < S;
    const uint32_t xoffN = (uint32_t)((more ? hn : hi) * p.W + (more ? wn : wi)) * p.Cin * (uint32_t)XROWB;
    if (KT > 0 && more) mfma_load<KTT>(ops, xbase, xoffN, p.ctrd, Cs, Cg, mgn * G, M, wave, lane);

    if (ccnt > 0) {
      const char* stage = lds + (s & 1) * STAGE_BYTES + lane * 8;
      const int mLast = min(M, (mg + 1) * G);
      for (int m = mg * G; m < mLast; ++m) {
        const uint32_t* __restrict__ tapBase = p.offs + m * p.Ct + c0;
#pragma unroll
        for (int dy = 0; dy < TH; ++dy) {
          const int kh = hi - rowStart[dy];
          if ((unsigned)kh < (unsigned)p.knl) {
#pragma unroll
            for (int dx = 0; dx < TW; ++dx) {
              const int kw = wi - colStart[dx];
              if ((unsigned)kw < (unsigned)p.knl)
                gather_row<CPW>(acc[dy * TW + dx], tapBase + (kh * p.knl + kw) * MCt, stage);
            }
          }
        }
      }
    }

    if (more) {
      char* nstage = lds + ((s + 1) & 1) * STAGE_BYTES;
      if (KT > 0) mfma_store<KTT>(ops, nstage, ksteps, Cs, Cg, mgn * G, M, wave, lane);
      else build_stage_exact(nstage, xbase, xoffN, p.ctrd, K, Cs, Cg, G, mgn * G, M, wave, lane);
    }
    __syncthreads();
    hi = hn; wi = wn; mg = mgn;
  }

  if (ccnt > 0) {
    float* __restrict__ dst = p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ho = ho0 + t / TW, wo = wo0 + t % TW;
      if (ho < p.Ho && wo < p.Wo) {
        float* o = dst + ((size_t)(ho * p.Wo + wo) * p.Ct + c0) * PANEL + 2 * lane;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          if (c < ccnt) {
            f32x2 v = acc[t][c];
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
// fully connected: stages of G sub-spaces, CPW channels per wave, optional split over the sub-space
// axis (blockIdx.z): partial sums go to p.partial and are reduced by k_sum_partials.
// ------------------------------------------------------------------------------------------------
template <int CPW, int KT>
__global__ __launch_bounds__(NW * 64) void k_fc_aprx(FcParams p, int G, int stagesPerSplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int KTT = KT > 0 ? KT : 1;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int panel = blockIdx.y;
  const int split = blockIdx.z;
  const int cw0 = blockIdx.x * (NW * CPW) + wave * CPW;
  const int ccnt = min(CPW, p.Ct - cw0);
  const int K = p.K, M = p.M, Cs = p.Cs;
  const int mBeg = split * stagesPerSplit * G;
  const int mEnd = min(M, mBeg + stagesPerSplit * G);
  const int S = (mEnd - mBeg + G - 1) / G;
  const char* __restrict__ xbase = reinterpret_cast<const char*>(p.src + (size_t)panel * p.D * PANEL);
  const int ksteps = (min(p.D, Cs) > 4) ? 2 : 1;

  f32x2 acc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float b = (split == 0) ? p.bias[cw0 + c] : 0.0f;   // over-read stays inside the arena
    acc[c] = f32x2{b, b};
  }

  MfmaOps<KTT> ops;
  if (S > 0) {
    if (KT > 0) {
      mfma_load<KTT>(ops, xbase, 0u, p.ctrd, Cs, p.D, mBeg, mEnd, wave, lane);
      mfma_store<KTT>(ops, lds, ksteps, Cs, p.D, mBeg, mEnd, wave, lane);
    } else {
      build_stage_exact(lds, xbase, 0u, p.ctrd, K, Cs, p.D, G, mBeg, mEnd, wave, lane);
    }
  }
  __syncthreads();

  for (int s = 0; s < S; ++s) {
    const int m0 = mBeg + s * G;
    const bool more = s + 1 < S;
    if (KT > 0 && more) mfma_load<KTT>(ops, xbase, 0u, p.ctrd, Cs, p.D, m0 + G, mEnd, wave, lane);

    if (ccnt > 0) {
      const char* stage = lds + (s & 1) * STAGE_BYTES + lane * 8;
      const int mLast = min(mEnd, m0 + G);
      for (int m = m0; m < mLast; ++m) gather_row<CPW>(acc, p.offs + (size_t)m * p.Ct + cw0, stage);
    }

    if (more) {
      char* nstage = lds + ((s + 1) & 1) * STAGE_BYTES;
      if (KT > 0) mfma_store<KTT>(ops, nstage, ksteps, Cs, p.D, m0 + G, mEnd, wave, lane);
      else build_stage_exact(nstage, xbase, 0u, p.ctrd, K, Cs, p.D, G, m0 + G, mEnd, wave, lane);
    }
    __syncthreads();
  }

  if (ccnt > 0) {
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

template <int TH, int TW, int CPW>
hipError_t launch_conv(const ConvParams& p, int lutMode, hipStream_t st) {
  const int Ctg = p.Ct / p.grp;
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH;
  const int chunksPerGrp = (Ctg + NW * CPW - 1) / (NW * CPW);
  const dim3 grid(tilesX * tilesY, chunksPerGrp * p.grp, p.panels);
  const size_t shm = (size_t)2 * STAGE_BYTES;
  const int G = qcnn_stage_group(p.K);
  auto kern = k_conv_aprx<TH, TW, CPW, 0>;
  if (lutMode == 1 && p.K == 128) kern = k_conv_aprx<TH, TW, CPW, 8>;
  if (lutMode == 1 && p.K == 64) kern = k_conv_aprx<TH, TW, CPW, 4>;
  if (lutMode == 1 && p.K == 32) kern = k_conv_aprx<TH, TW, CPW, 2>;
  if (lutMode == 1 && p.K == 16) kern = k_conv_aprx<TH, TW, CPW, 1>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, tilesX, chunksPerGrp, G);
  return hipGetLastError();
}

template <int CPW>
hipError_t launch_fc(const FcParams& p, int lutMode, hipStream_t st) {
  const int G = qcnn_stage_group(p.K);
  const int stages = (p.M + G - 1) / G;
  const int stagesPerSplit = (stages + p.msplit - 1) / p.msplit;
  const dim3 grid((p.Ct + NW * CPW - 1) / (NW * CPW), p.panels, p.msplit);
  const size_t shm = (size_t)2 * STAGE_BYTES;
  auto kern = k_fc_aprx<CPW, 0>;
  if (lutMode == 1 && p.K == 128) kern = k_fc_aprx<CPW, 8>;
  if (lutMode == 1 && p.K == 64) kern = k_fc_aprx<CPW, 4>;
  if (lutMode == 1 && p.K == 32) kern = k_fc_aprx<CPW, 2>;
  if (lutMode == 1 && p.K == 16) kern = k_fc_aprx<CPW, 1>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, G, stagesPerSplit);
  return hipGetLastError();
}

}  // namespace

// Tile selection: 8 waves; channels-per-wave from the group's channel count; the position tile is as
// large as ~72 float2 accumulators per lane allow (more starve the LDS-read pipeline of registers).
// The MFMA builder is instantiated for K in {16, 32, 64, 128}; any other K <= 128 runs the exact builder.
hipError_t qk_conv_aprx(const ConvParams& p, int lutMode, hipStream_t st) {
  const int Ctg = p.Ct / p.grp;
  if (Ctg % 4 || p.Cs > QCNN_MAX_CS || p.K > QCNN_MAX_K) return hipErrorInvalidValue;
  if (Ctg % 384 == 0) return launch_conv<1, 1, 48>(p, lutMode, st);
  if (Ctg % 256 == 0) return launch_conv<1, 2, 32>(p, lutMode, st);
  if (Ctg % 192 == 0) return launch_conv<1, 3, 24>(p, lutMode, st);
  if (Ctg % 128 == 0) return launch_conv<2, 2, 16>(p, lutMode, st);
  if (Ctg % 96 == 0) return launch_conv<2, 3, 12>(p, lutMode, st);
  if (Ctg % 64 == 0) return launch_conv<2, 4, 8>(p, lutMode, st);
  if (Ctg > 64) return launch_conv<2, 2, 16>(p, lutMode, st);
  return launch_conv<3, 4, 4>(p, lutMode, st);
}

// p.msplit is chosen by the caller (engine): 1 keeps the reference's summation order.
hipError_t qk_fc_aprx(const FcParams& p, int lutMode, hipStream_t st) {
  if (p.Ct % 4 || p.Cs > QCNN_MAX_CS || p.K > QCNN_MAX_K || p.msplit < 1) return hipErrorInvalidValue;
  if (p.Ct >= 2048) return launch_fc<64>(p, lutMode, st);
  if (p.Ct >= 512) return launch_fc<32>(p, lutMode, st);
  if (p.Ct >= 64) return launch_fc<8>(p, lutMode, st);
  return launch_fc<4>(p, lutMode, st);
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
  const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_lrn, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, rows, C, lrnSiz, coeff, -bet, ini);
  return hipGetLastError();
}

hipError_t qk_pool(const float* src, float* dst, int panels, int H, int W, int C, int Ho, int Wo, int knl, int stride,
                   int pad, hipStream_t st) {
  const size_t rows = (size_t)panels * Ho * Wo * C;
  const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_pool, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, panels, H, W, C, Ho, Wo, knl,
                     stride, pad);
  return hipGetLastError();
}

hipError_t qk_softmax(const float* src, float* dst, int panels, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_softmax, dim3((panels * PANEL + 63) / 64), dim3(64), 0, st, src, dst, panels, C);
  return hipGetLastError();
}

hipError_t qk_top5(const float* prob, uint16_t* out, int n, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_top5, dim3((n + 63) / 64), dim3(64), 0, st, prob, out, n, C);
  return hipGetLastError();
}

hipError_t qk_pack_nchw(const float* in, float* dst, int n, int C, int H, int W, hipStream_t st) {
  const int E = C * H * W;
  hipLaunchKernelGGL(k_pack, dim3((E + 63) / 64, panels_of(n)), dim3(256), 0, st, in, dst, n, E, C, H * W, 1);
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
