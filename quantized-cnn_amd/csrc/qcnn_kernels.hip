// qcnn_kernels.hip — hand-written gfx950 (CDNA4) kernels of the Quantized-CNN approximate forward pass.
//
// Hot kernels:
//   k_conv_aprx  fused  GetInPdMat (src/CaffeEva.cc:1261-1296)  +  CalcFeatMap_ConvAprx (:760-868)
//   k_fc_aprx    fused  GetInPdMat                              +  CalcFeatMap_FCntAprx (:968-1025)
// Glue kernels (row a9): ReLU :1027, LRN :1038, max-pool :870, softmax :1098, top-5 :1162,
// NCHW<->panel conversions (:1146-1160, :187-189).
//
// Mapping (see qcnn_kernels.h for the HBM layout): a lane carries an image pair.  A workgroup owns one
// 128-image panel, one tile of output positions and one slice of output channels; every wave keeps
// (positions x channels-per-wave) float2 accumulators in VGPRs.  The look-up table is never
// materialised in HBM: it is produced one "slot" at a time in LDS — slot(p, m) = the K inner products of
// sub-space m of source pixel p for the 128 images, laid out [K][128] so that a code-word row is 512
// contiguous bytes = one conflict-free ds_read_b64 per wave — double buffered, built by all waves
// (v_mfma_f32_16x16x4_f32 with operands prefetched one slot ahead, or ordered VALU mul+add in "exact"
// mode), then consumed by every (position, channel) of the tile whose receptive field contains p.
// Slots are visited in (pixel row-major, m ascending) order, which for any one output is exactly the
// reference's (kh, kw, m) summation order (:840-863), so with the exact builder conv/FC outputs are
// bit-identical to the reference.  Code-word offsets are wave-uniform and come in through scalar loads.
#include "qcnn_kernels.h"

#include <float.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PANEL = QCNN_PANEL;          // images per panel
constexpr int ROWB = PANEL * 4;            // bytes of one code-word row of a slot

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// CPW look-ups of one (position, sub-space): the code-word offsets are wave-uniform and arrive four at a
// time through s_load_dwordx4 (the tables are 16-byte aligned: Ct and the wave's first channel are
// multiples of 4); each look-up is one ds_read_b64 of the image pair + one v_pk_add_f32.
template <int CPW>
__device__ __forceinline__ void gather_row(f32x2 (&acc)[CPW], const uint32_t* __restrict__ ap, const char* slot) {
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
      v[4 * j + 0] = *reinterpret_cast<const f32x2*>(slot + o[j].x);
      v[4 * j + 1] = *reinterpret_cast<const f32x2*>(slot + o[j].y);
      v[4 * j + 2] = *reinterpret_cast<const f32x2*>(slot + o[j].z);
      v[4 * j + 3] = *reinterpret_cast<const f32x2*>(slot + o[j].w);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) acc[g0 + j] += v[j];
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------------------
// LUT slot builders.  slot: LDS [K][128] floats.  xrow(d): global pointer to the 128-image row of
// input dim d of this sub-space.  ctrdM: code book of sub-space m, [Cs][K].  dsel = dims that exist.
// ------------------------------------------------------------------------------------------------

// exact: y = ((0 + x0*c0) + x1*c1) + ...  with separately rounded product and sum, the order of the
// reference's saxpy chain (src/CaffeEva.cc:1284-1289, include/BlasWrapper.h:164-184).
template <int NW, typename RowFn>
__device__ __forceinline__ void build_slot_exact(float* slot, const float* __restrict__ ctrdM, int K, int dsel,
                                                 int wave, int lane, RowFn xrow) {
  f32x2 xv[QCNN_MAX_CS];
#pragma unroll
  for (int d = 0; d < QCNN_MAX_CS; ++d) {
    xv[d] = f32x2{0.0f, 0.0f};
    if (d < dsel) xv[d] = *reinterpret_cast<const f32x2*>(xrow(d) + 2 * lane);
  }
  const int kpw = (K + NW - 1) / NW;
  const int k0 = wave * kpw;
  const int k1 = min(K, k0 + kpw);
  for (int k = k0; k < k1; ++k) {
    float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
    for (int d = 0; d < QCNN_MAX_CS; ++d) {
      if (d < dsel) {
        const float c = ctrdM[d * K + k];
        v0 = __fadd_rn(v0, __fmul_rn(xv[d].x, c));
        v1 = __fadd_rn(v1, __fmul_rn(xv[d].y, c));
      }
    }
    *reinterpret_cast<f32x2*>(slot + k * PANEL + 2 * lane) = f32x2{v0, v1};
  }
}

// MFMA: D[16 code words][16 images] += A[16 code words x 4 dims] * B[4 dims x 16 images]
// (v_mfma_f32_16x16x4_f32).  Lane l holds A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)*4 + r][l&15].
// A slot is KT x 8 tiles (KT = K/16 code-word tiles, 8 image tiles); wave w owns tiles w, w+NW, ...
// Operands are fetched into registers (mfma_load) well before they are consumed (mfma_store).
template <int KT, int NW>
struct MfmaOps {
  static constexpr int TPW = (KT * 8 + NW - 1) / NW;      // tiles per wave
  static constexpr int XT = (NW % 8 == 0) ? 1 : TPW;      // with NW % 8 == 0 a wave keeps one image tile
  float a[TPW][2];   // code-book operand per tile and k-step
  float b[XT][2];    // input operand per image tile and k-step
};

template <int KT, int NW, typename RowFn>
__device__ __forceinline__ void mfma_load(MfmaOps<KT, NW>& o, const float* __restrict__ ctrdM, int dsel, int wave,
                                          int lane, RowFn xrow) {
  constexpr int K = KT * 16;
  using Ops = MfmaOps<KT, NW>;
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int d = ks * 4 + lk;
    const bool dv = d < dsel;
#pragma unroll
    for (int i = 0; i < Ops::XT; ++i) {
      const int it = (wave + NW * i) & 7;
      o.b[i][ks] = dv ? xrow(d)[it * 16 + li] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < Ops::TPW; ++i) {
      const int q = wave + NW * i;
      o.a[i][ks] = (dv && q < KT * 8) ? ctrdM[d * K + (q >> 3) * 16 + li] : 0.0f;
    }
  }
}

template <int KT, int NW>
__device__ __forceinline__ void mfma_store(const MfmaOps<KT, NW>& o, float* slot, int dsel, int wave, int lane) {
  using Ops = MfmaOps<KT, NW>;
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int i = 0; i < Ops::TPW; ++i) {
    const int q = wave + NW * i;
    if (q < KT * 8) {
      const int kt = q >> 3, it = q & 7;
      const int xi = (Ops::XT == 1) ? 0 : i;
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][0], o.b[xi][0], acc, 0, 0, 0);
      if (dsel > 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i][1], o.b[xi][1], acc, 0, 0, 0);
      float* w = slot + (kt * 16 + lk * 4) * PANEL + it * 16 + li;
      w[0] = acc[0];
      w[PANEL] = acc[1];
      w[2 * PANEL] = acc[2];
      w[3 * PANEL] = acc[3];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv: TH x TW output positions, NW waves, CPW channels per wave; KT = K/16 for the MFMA builder,
// KT = 0 selects the exact builder (any K).
// ------------------------------------------------------------------------------------------------
template <int TH, int TW, int CPW, int NW, int KT>
__global__ __launch_bounds__(NW * 64) void k_conv_aprx(ConvParams p, int tilesX, int chunksPerGrp) {
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
  const int ccnt = min(CPW, Ctg - cw0);              // <= 0: the wave only helps building slots
  const int c0 = g * Ctg + cw0;
  const int K = p.K, M = p.M, Cs = p.Cs;
  const int slotBytes = K * ROWB;

  const float* __restrict__ src = p.src + (size_t)panel * p.H * p.W * p.Cin * PANEL;
  const int chanBase = g * Cg;

  f32x2 acc[NT][CPW];
  {
    const float* __restrict__ bp = p.bias + c0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const float b = (c < ccnt) ? bp[c] : 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][c] = f32x2{b, b};
    }
  }

  const int ho0 = ty * TH, wo0 = tx * TW;
  const int hoL = min(ho0 + TH, p.Ho) - 1, woL = min(wo0 + TW, p.Wo) - 1;   // last real position of the tile
  const int hiL = max(0, ho0 * p.stride - p.pad), hiU = min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1);
  const int wiL = max(0, wo0 * p.stride - p.pad), wiU = min(p.W - 1, woL * p.stride - p.pad + p.knl - 1);
  const int S = (hiU - hiL + 1) * (wiU - wiL + 1) * M;

  // first source row / column of every position of the tile; positions outside the map get a start
  // that can never match a tap
  int rowStart[TH], colStart[TW];
#pragma unroll
  for (int dy = 0; dy < TH; ++dy) rowStart[dy] = (ho0 + dy < p.Ho) ? (ho0 + dy) * p.stride - p.pad : -(1 << 28);
#pragma unroll
  for (int dx = 0; dx < TW; ++dx) colStart[dx] = (wo0 + dx < p.Wo) ? (wo0 + dx) * p.stride - p.pad : -(1 << 28);

  auto xptr = [&](int hi, int wi, int m) {
    return src + ((size_t)(hi * p.W + wi) * p.Cin + chanBase + m * Cs) * PANEL;
  };

  MfmaOps<KTT, NW> ops;
  int hi = hiL, wi = wiL, m = 0;
  {
    const float* __restrict__ xp = xptr(hi, wi, m);
    const float* __restrict__ cm = p.ctrd + (size_t)m * Cs * K;
    const int dsel = min(Cg - m * Cs, Cs);
    auto xrow = [&](int d) { return xp + d * PANEL; };
    if (KT > 0) {
      mfma_load<KTT, NW>(ops, cm, dsel, wave, lane, xrow);
      mfma_store<KTT, NW>(ops, reinterpret_cast<float*>(lds), dsel, wave, lane);
    } else {
      build_slot_exact<NW>(reinterpret_cast<float*>(lds), cm, K, dsel, wave, lane, xrow);
    }
  }
  __syncthreads();

  for (int s = 0; s < S; ++s) {
    int mn = m + 1, wn = wi, hn = hi;
    if (mn == M) {
      mn = 0;
      if (++wn > wiU) { wn = wiL; ++hn; }
    }
    const bool more = s + 1 < S;
    const int dselN = min(Cg - mn * Cs, Cs);
    const float* __restrict__ xpN = xptr(more ? hn : hi, more ? wn : wi, more ? mn : m);
    const float* __restrict__ cmN = p.ctrd + (size_t)mn * Cs * K;
    auto xrowN = [&](int d) { return xpN + d * PANEL; };
    if (KT > 0 && more) mfma_load<KTT, NW>(ops, cmN, dselN, wave, lane, xrowN);

    if (ccnt > 0) {
      const char* slot = lds + (s & 1) * slotBytes + lane * 8;
      const uint32_t* __restrict__ tapBase = p.offs + (size_t)m * p.Ct + c0;
#pragma unroll
      for (int dy = 0; dy < TH; ++dy) {
        const int kh = hi - rowStart[dy];
        if ((unsigned)kh < (unsigned)p.knl) {
#pragma unroll
          for (int dx = 0; dx < TW; ++dx) {
            const int kw = wi - colStart[dx];
            if ((unsigned)kw < (unsigned)p.knl) {
              const uint32_t* __restrict__ ap = tapBase + (size_t)(kh * p.knl + kw) * M * p.Ct;
              gather_row<CPW>(acc[dy * TW + dx], ap, slot);
            }
          }
        }
      }
    }

    if (more) {
      float* nslot = reinterpret_cast<float*>(lds + ((s + 1) & 1) * slotBytes);
      if (KT > 0) mfma_store<KTT, NW>(ops, nslot, dselN, wave, lane);
      else build_slot_exact<NW>(nslot, cmN, K, dselN, wave, lane, xrowN);
    }
    __syncthreads();
    hi = hn; wi = wn; m = mn;
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
// fully connected: slot(m) for the panel, CPW channels per wave
// ------------------------------------------------------------------------------------------------
template <int CPW, int NW, int KT>
__global__ __launch_bounds__(NW * 64) void k_fc_aprx(FcParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int KTT = KT > 0 ? KT : 1;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int panel = blockIdx.y;
  const int cw0 = blockIdx.x * (NW * CPW) + wave * CPW;
  const int ccnt = min(CPW, p.Ct - cw0);
  const int K = p.K, M = p.M, Cs = p.Cs;
  const int slotBytes = K * ROWB;
  const float* __restrict__ src = p.src + (size_t)panel * p.D * PANEL;
  const int* __restrict__ dmap = p.dmap;

  f32x2 acc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float b = (c < ccnt) ? p.bias[cw0 + c] : 0.0f;
    acc[c] = f32x2{b, b};
  }

  MfmaOps<KTT, NW> ops;
  auto xrow_of = [&](int m) {
    return [=](int d) {
      const int e = m * Cs + d;
      const int row = dmap ? dmap[e] : e;
      return src + (size_t)row * PANEL;
    };
  };
  {
    const int dsel = min(p.D, Cs);
    auto xrow = xrow_of(0);
    if (KT > 0) {
      mfma_load<KTT, NW>(ops, p.ctrd, dsel, wave, lane, xrow);
      mfma_store<KTT, NW>(ops, reinterpret_cast<float*>(lds), dsel, wave, lane);
    } else {
      build_slot_exact<NW>(reinterpret_cast<float*>(lds), p.ctrd, K, dsel, wave, lane, xrow);
    }
  }
  __syncthreads();

  for (int m = 0; m < M; ++m) {
    const bool more = m + 1 < M;
    const int mn = more ? m + 1 : m;
    const int dselN = min(p.D - mn * Cs, Cs);
    const float* __restrict__ cmN = p.ctrd + (size_t)mn * Cs * K;
    auto xrowN = xrow_of(mn);
    if (KT > 0 && more) mfma_load<KTT, NW>(ops, cmN, dselN, wave, lane, xrowN);

    if (ccnt > 0) {
      const char* slot = lds + (m & 1) * slotBytes + lane * 8;
      const uint32_t* __restrict__ ap = p.offs + (size_t)m * p.Ct + cw0;
      gather_row<CPW>(acc, ap, slot);
    }

    if (more) {
      float* nslot = reinterpret_cast<float*>(lds + ((m + 1) & 1) * slotBytes);
      if (KT > 0) mfma_store<KTT, NW>(ops, nslot, dselN, wave, lane);
      else build_slot_exact<NW>(nslot, cmN, K, dselN, wave, lane, xrowN);
    }
    __syncthreads();
  }

  if (ccnt > 0) {
    float* o = p.dst + ((size_t)panel * p.Ct + cw0) * PANEL + 2 * lane;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (c < ccnt) {
        f32x2 v = acc[c];
        if (p.relu) {
          v.x = (0.0f < v.x) ? v.x : 0.0f;
          v.y = (0.0f < v.y) ? v.y : 0.0f;
        }
        *reinterpret_cast<f32x2*>(o + c * PANEL) = v;
      }
    }
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

template <int TH, int TW, int CPW, int NW>
hipError_t launch_conv(const ConvParams& p, int lutMode, hipStream_t st) {
  const int Ctg = p.Ct / p.grp;
  const int tilesX = (p.Wo + TW - 1) / TW, tilesY = (p.Ho + TH - 1) / TH;
  const int chunksPerGrp = (Ctg + NW * CPW - 1) / (NW * CPW);
  const dim3 grid(tilesX * tilesY, chunksPerGrp * p.grp, p.panels);
  const size_t shm = (size_t)2 * p.K * ROWB;
  auto kern = (lutMode == 1 && p.K == 128) ? k_conv_aprx<TH, TW, CPW, NW, 8> : k_conv_aprx<TH, TW, CPW, NW, 0>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, tilesX, chunksPerGrp);
  return hipGetLastError();
}

template <int CPW, int NW>
hipError_t launch_fc(const FcParams& p, int lutMode, hipStream_t st) {
  const dim3 grid((p.Ct + NW * CPW - 1) / (NW * CPW), p.panels);
  const size_t shm = (size_t)2 * p.K * ROWB;
  auto kern = k_fc_aprx<CPW, NW, 0>;
  if (lutMode == 1 && p.K == 32) kern = k_fc_aprx<CPW, NW, 2>;
  if (lutMode == 1 && p.K == 16) kern = k_fc_aprx<CPW, NW, 1>;
  if (lutMode == 1 && p.K == 128) kern = k_fc_aprx<CPW, NW, 8>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p);
  return hipGetLastError();
}

}  // namespace

// Tile selection: 8 waves; channels-per-wave from the group's channel count; the position tile is as
// large as ~72 float2 accumulators per lane allow (more starve the LDS-read pipeline of registers) (more positions per tile = more reuse of a LUT slot).
// The MFMA builder is instantiated for K = 128 (conv) and K = 16 / 32 / 128 (FC); any other K runs
// the exact builder.
hipError_t qk_conv_aprx(const ConvParams& p, int lutMode, hipStream_t st) {
  const int Ctg = p.Ct / p.grp;
  if (Ctg % 4 || p.Cs > QCNN_MAX_CS || p.K > 256 || (size_t)2 * p.K * ROWB > 160 * 1024) return hipErrorInvalidValue;
  if (Ctg % 384 == 0) return launch_conv<1, 1, 48, 8>(p, lutMode, st);
  if (Ctg % 256 == 0) return launch_conv<1, 2, 32, 8>(p, lutMode, st);
  if (Ctg % 192 == 0) return launch_conv<1, 3, 24, 8>(p, lutMode, st);
  if (Ctg % 128 == 0) return launch_conv<2, 2, 16, 8>(p, lutMode, st);
  if (Ctg % 96 == 0) return launch_conv<2, 3, 12, 8>(p, lutMode, st);
  if (Ctg % 64 == 0) return launch_conv<2, 4, 8, 8>(p, lutMode, st);
  if (Ctg > 64) return launch_conv<2, 2, 16, 8>(p, lutMode, st);
  return launch_conv<3, 4, 4, 8>(p, lutMode, st);
}

hipError_t qk_fc_aprx(const FcParams& p, int lutMode, hipStream_t st) {
  if (p.Ct % 4 || p.Cs > QCNN_MAX_CS || p.K > 256 || (size_t)2 * p.K * ROWB > 160 * 1024) return hipErrorInvalidValue;
  if (p.Ct >= 2048) return launch_fc<32, 4>(p, lutMode, st);
  if (p.Ct >= 256) return launch_fc<16, 4>(p, lutMode, st);
  return launch_fc<4, 4>(p, lutMode, st);
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
