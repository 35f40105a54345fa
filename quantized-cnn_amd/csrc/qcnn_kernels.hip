// qcnn_kernels.hip — hand-written gfx950 (CDNA4) kernels of the Quantized-CNN approximate forward pass.
//
// Hot kernels:
//   k_conv_aprx  fused  GetInPdMat (src/CaffeEva.cc:1261-1296)  +  CalcFeatMap_ConvAprx (:760-868)
//   k_fc_aprx    fused  GetInPdMat                              +  CalcFeatMap_FCntAprx (:968-1025)
// Glue kernels (row a9): ReLU :1027, LRN :1038, max-pool :870, softmax :1098, top-5 :1162,
// NCHW<->panel conversions (:1146-1160, :187-189).
//
// Mapping (see qcnn_kernels.h for the HBM layout): lane = image.  A workgroup owns one 64-image panel,
// one tile of output positions and one slice of output channels; every wave keeps
// (positions x channels-per-wave) fp32 accumulators in VGPRs.  The look-up table is never
// materialised in HBM: it is produced one "slot" at a time in LDS — slot(p, m) = the K inner
// products of sub-space m of source pixel p for the 64 images, laid out [K][64] so that a code-word
// row is 256 contiguous bytes = one conflict-free ds_read_b32 per wave — double buffered, built by
// all waves (MFMA v_mfma_f32_16x16x4_f32, or ordered VALU mul+add in "exact" mode), then consumed by
// every (position, channel) of the tile whose receptive field contains p.  Slots are visited in
// (pixel row-major, m ascending) order, which for any one output is exactly the reference's
// (kh, kw, m) summation order (:840-863), so with the exact builder conv/FC outputs are bit-identical
// to the reference.  Assignment indices are wave-uniform and come in through scalar loads.
#include "qcnn_kernels.h"

#include <float.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PANEL = QCNN_PANEL;

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ------------------------------------------------------------------------------------------------
// LUT slot builders.  slot: LDS [K][64].  xrow(d): pointer to the 64-image row of input dim d of
// this sub-space.  ctrdM: code book of sub-space m, [Cs][K].  dsel = dims that exist (CsEff).
// ------------------------------------------------------------------------------------------------

// exact: y = ((0 + x0*c0) + x1*c1) + ...  with separately rounded product and sum, the order of the
// reference's saxpy chain (src/CaffeEva.cc:1284-1289, include/BlasWrapper.h:164-184).
template <int NW, typename RowFn>
__device__ __forceinline__ void build_slot_exact(float* slot, const float* __restrict__ ctrdM, int K, int dsel,
                                                 int wave, int lane, RowFn xrow) {
  float xv[QCNN_MAX_CS];
#pragma unroll
  for (int d = 0; d < QCNN_MAX_CS; ++d) xv[d] = (d < dsel) ? xrow(d)[lane] : 0.0f;
  const int kpw = (K + NW - 1) / NW;
  const int k0 = wave * kpw;
  const int k1 = min(K, k0 + kpw);
  for (int k = k0; k < k1; ++k) {
    float v = 0.0f;
#pragma unroll
    for (int d = 0; d < QCNN_MAX_CS; ++d) {
      if (d < dsel) v = __fadd_rn(v, __fmul_rn(xv[d], ctrdM[d * K + k]));
    }
    slot[k * PANEL + lane] = v;
  }
}

// MFMA: D[16 code words][16 images] += A[16 x 4 dims] * B[4 dims x 16 images], v_mfma_f32_16x16x4_f32.
// A lane l holds A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)*4 + r][l&15].  K % 16 == 0 required.
template <int NW, typename RowFn>
__device__ __forceinline__ void build_slot_mfma(float* slot, const float* __restrict__ ctrdM, int K, int Cs, int dsel,
                                                int wave, int lane, RowFn xrow) {
  const int tiles = (K >> 4) * (PANEL / 16);
  const int li = lane & 15, lk = lane >> 4;
  for (int q = wave; q < tiles; q += NW) {
    const int kt = q >> 2, jt = q & 3;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int d0 = 0; d0 < dsel; d0 += 4) {
      const int d = d0 + lk;
      float a = 0.0f, b = 0.0f;
      if (d < dsel) {
        a = ctrdM[d * K + kt * 16 + li];
        b = xrow(d)[jt * 16 + li];
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    float* o = slot + (kt * 16 + lk * 4) * PANEL + jt * 16 + li;
    o[0] = acc[0];
    o[PANEL] = acc[1];
    o[2 * PANEL] = acc[2];
    o[3 * PANEL] = acc[3];
  }
  (void)Cs;
}

// ------------------------------------------------------------------------------------------------
// conv: TH x TW output positions, NW waves, CPW channels per wave
// ------------------------------------------------------------------------------------------------
template <int TH, int TW, int CPW, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void k_conv_aprx(ConvParams p, int tilesX, int chunksPerGrp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(CPW % 4 == 0, "assignment words are fetched as uint32");
  constexpr int NT = TH * TW;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int ty = blockIdx.x / tilesX, tx = blockIdx.x % tilesX;
  const int g = blockIdx.y / chunksPerGrp, chunk = blockIdx.y % chunksPerGrp;
  const int panel = blockIdx.z;
  const int Cg = p.Cin / p.grp, Ctg = p.Ct / p.grp;
  const int cw0 = chunk * (NW * CPW) + wave * CPW;   // first channel of this wave inside the group
  const int ccnt = min(CPW, Ctg - cw0);              // <= 0: wave only helps building slots
  const int c0 = g * Ctg + cw0;
  const int K = p.K, M = p.M, Cs = p.Cs;
  const int slotElems = K * PANEL;

  const float* __restrict__ src = p.src + (size_t)panel * p.H * p.W * p.Cin * PANEL;
  const int chanBase = g * Cg;

  float acc[NT][CPW];
  {
    const float* __restrict__ bp = p.bias + c0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const float b = (c < ccnt) ? bp[c] : 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][c] = b;
    }
  }

  const int ho0 = ty * TH, wo0 = tx * TW;
  const int hoL = min(ho0 + TH, p.Ho) - 1, woL = min(wo0 + TW, p.Wo) - 1;   // last real position of the tile
  const int hiL = max(0, ho0 * p.stride - p.pad), hiU = min(p.H - 1, hoL * p.stride - p.pad + p.knl - 1);
  const int wiL = max(0, wo0 * p.stride - p.pad), wiU = min(p.W - 1, woL * p.stride - p.pad + p.knl - 1);
  const int nw = wiU - wiL + 1;
  const int S = (hiU - hiL + 1) * nw * M;

  auto build = [&](int s, float* slot) {
    const int m = s % M, pix = s / M;
    const int hi = hiL + pix / nw, wi = wiL + pix % nw;
    const int dsel = min(Cg - m * Cs, Cs);
    const float* __restrict__ xp = src + ((size_t)(hi * p.W + wi) * p.Cin + chanBase + m * Cs) * PANEL;
    const float* __restrict__ cm = p.ctrd + (size_t)m * Cs * K;
    auto xrow = [&](int d) { return xp + d * PANEL; };
    if (MODE == 0) build_slot_exact<NW>(slot, cm, K, dsel, wave, lane, xrow);
    else build_slot_mfma<NW>(slot, cm, K, Cs, dsel, wave, lane, xrow);
  };

  build(0, lds);
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    if (s + 1 < S) build(s + 1, lds + ((s + 1) & 1) * slotElems);
    if (ccnt > 0) {
      const int m = s % M, pix = s / M;
      const int hi = hiL + pix / nw, wi = wiL + pix % nw;
      const float* slot = lds + (s & 1) * slotElems + lane;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int ho = ho0 + t / TW, wo = wo0 + t % TW;
        const int kh = hi - (ho * p.stride - p.pad), kw = wi - (wo * p.stride - p.pad);
        if (kh >= 0 && kh < p.knl && kw >= 0 && kw < p.knl && ho < p.Ho && wo < p.Wo) {
          const uint32_t* __restrict__ ap =
              reinterpret_cast<const uint32_t*>(p.asmt + ((size_t)(kh * p.knl + kw) * M + m) * p.Ct + c0);
          uint32_t wv[CPW / 4];
#pragma unroll
          for (int j = 0; j < CPW / 4; ++j) wv[j] = ap[j];
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const uint32_t idx = (wv[c >> 2] >> ((c & 3) * 8)) & 0xffu;
            acc[t][c] += slot[idx * PANEL];
          }
        }
      }
    }
    __syncthreads();
  }

  if (ccnt > 0) {
    float* __restrict__ dst = p.dst + (size_t)panel * p.Ho * p.Wo * p.Ct * PANEL;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ho = ho0 + t / TW, wo = wo0 + t % TW;
      if (ho < p.Ho && wo < p.Wo) {
        float* o = dst + ((size_t)(ho * p.Wo + wo) * p.Ct + c0) * PANEL + lane;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          if (c < ccnt) {
            float v = acc[t][c];
            if (p.relu) v = (0.0f < v) ? v : 0.0f;
            o[c * PANEL] = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fully connected: slot(m) for the panel, CPW channels per wave
// ------------------------------------------------------------------------------------------------
template <int CPW, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void k_fc_aprx(FcParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(CPW % 4 == 0, "assignment words are fetched as uint32");
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int panel = blockIdx.y;
  const int cw0 = blockIdx.x * (NW * CPW) + wave * CPW;
  const int ccnt = min(CPW, p.Ct - cw0);
  const int K = p.K, M = p.M, Cs = p.Cs;
  const int slotElems = K * PANEL;
  const float* __restrict__ src = p.src + (size_t)panel * p.D * PANEL;
  const int* __restrict__ dmap = p.dmap;

  float acc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) acc[c] = (c < ccnt) ? p.bias[cw0 + c] : 0.0f;

  auto build = [&](int m, float* slot) {
    const int dsel = min(p.D - m * Cs, Cs);
    const float* __restrict__ cm = p.ctrd + (size_t)m * Cs * K;
    auto xrow = [&](int d) {
      const int e = m * Cs + d;
      const int row = dmap ? dmap[e] : e;
      return src + (size_t)row * PANEL;
    };
    if (MODE == 0) build_slot_exact<NW>(slot, cm, K, dsel, wave, lane, xrow);
    else build_slot_mfma<NW>(slot, cm, K, Cs, dsel, wave, lane, xrow);
  };

  build(0, lds);
  __syncthreads();
  for (int m = 0; m < M; ++m) {
    if (m + 1 < M) build(m + 1, lds + ((m + 1) & 1) * slotElems);
    if (ccnt > 0) {
      const float* slot = lds + (m & 1) * slotElems + lane;
      const uint32_t* __restrict__ ap = reinterpret_cast<const uint32_t*>(p.asmt + (size_t)m * p.Ct + cw0);
      uint32_t wv[CPW / 4];
#pragma unroll
      for (int j = 0; j < CPW / 4; ++j) wv[j] = ap[j];
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const uint32_t idx = (wv[c >> 2] >> ((c & 3) * 8)) & 0xffu;
        acc[c] += slot[idx * PANEL];
      }
    }
    __syncthreads();
  }

  if (ccnt > 0) {
    float* o = p.dst + ((size_t)panel * p.Ct + cw0) * PANEL + lane;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      if (c < ccnt) {
        float v = acc[c];
        if (p.relu) v = (0.0f < v) ? v : 0.0f;
        o[c * PANEL] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// glue kernels (all: lane = image, rows of 64)
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
    const float* x = src + r * PANEL + lane;
    float s = ini;
    for (int j = 0; j < lrnSiz; ++j) {
      const int cc = c - rad + j;
      if (cc >= 0 && cc < C) {
        const float xv = x[(ptrdiff_t)(cc - c) * PANEL];
        s = __fadd_rn(s, __fmul_rn(__fmul_rn(xv, xv), coeff));
      }
    }
    const float xc = x[0];
    dst[r * PANEL + lane] = __fmul_rn(xc, expf(__fmul_rn(nbet, logf(s))));
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
    const float* base = src + (size_t)panel * H * W * C * PANEL + lane;
    float v = 0.0f;
    bool first = true;
    for (int h = hL; h <= hU; ++h)
      for (int w = wL; w <= wU; ++w) {
        const float s = base[((size_t)(h * W + w) * C + c) * PANEL];
        v = first ? s : ((s < v) ? v : s);
        first = false;
      }
    dst[r * PANEL + lane] = v;
  }
}

// src/CaffeEva.cc:1098-1116: y = expf(x); sequential float sum over classes; y /= sum.  One thread = one image.
__global__ void k_softmax(const float* __restrict__ src, float* __restrict__ dst, int panels, int C) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= panels * PANEL) return;
  const int panel = t >> 6, lane = t & 63;
  const float* x = src + (size_t)panel * C * PANEL + lane;
  float* y = dst + (size_t)panel * C * PANEL + lane;
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
  const float* x = prob + (size_t)(img >> 6) * C * PANEL + (img & 63);
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

// [n][E] rows -> panels [E][64] through a 64x64 LDS tile (both sides coalesced).
// NCHW: input element e = (c*H + h)*W + w of an image lands in row (h*W + w)*C + c (src/CaffeEva.cc:1146-1160).
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ in, float* __restrict__ dst, int n, int E,
                                              int C, int HW, int nchw) {
  __shared__ float tile[64][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e0 = blockIdx.x * 64;
  const int panel = blockIdx.y;
  for (int i = wave; i < 64; i += 4) {
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
      dst[((size_t)panel * E + row) * PANEL + lane] = tile[lane][j];
    }
  }
}

// panels [E][64] -> [n][E]
__global__ __launch_bounds__(256) void k_unpack(const float* __restrict__ src, float* __restrict__ out, int n, int E) {
  __shared__ float tile[64][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e0 = blockIdx.x * 64;
  const int panel = blockIdx.y;
  for (int j = wave; j < 64; j += 4) {
    const int e = e0 + j;
    tile[j][lane] = (e < E) ? src[((size_t)panel * E + e) * PANEL + lane] : 0.0f;
  }
  __syncthreads();
  for (int i = wave; i < 64; i += 4) {
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
  const size_t shm = (size_t)2 * p.K * PANEL * sizeof(float);
  const bool mfma = lutMode == 1 && (p.K % 16) == 0;
  auto kern = mfma ? k_conv_aprx<TH, TW, CPW, NW, 1> : k_conv_aprx<TH, TW, CPW, NW, 0>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p, tilesX, chunksPerGrp);
  return hipGetLastError();
}

template <int CPW, int NW>
hipError_t launch_fc(const FcParams& p, int lutMode, hipStream_t st) {
  const dim3 grid((p.Ct + NW * CPW - 1) / (NW * CPW), p.panels);
  const size_t shm = (size_t)2 * p.K * PANEL * sizeof(float);
  const bool mfma = lutMode == 1 && (p.K % 16) == 0;
  auto kern = mfma ? k_fc_aprx<CPW, NW, 1> : k_fc_aprx<CPW, NW, 0>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)shm);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), shm, st, p);
  return hipGetLastError();
}

}  // namespace

// Tile selection: 8 waves; channels-per-wave from the group's channel count; the position tile is
// as large as ~192 accumulators per lane allow (more positions per tile = more reuse of a LUT slot).
hipError_t qk_conv_aprx(const ConvParams& p, int lutMode, hipStream_t st) {
  const int Ctg = p.Ct / p.grp;
  if (Ctg % 4 || p.Cs > QCNN_MAX_CS || p.K > 256) return hipErrorInvalidValue;
  if (Ctg % 384 == 0) return launch_conv<2, 2, 48, 8>(p, lutMode, st);
  if (Ctg % 256 == 0) return launch_conv<2, 3, 32, 8>(p, lutMode, st);
  if (Ctg % 192 == 0) return launch_conv<2, 4, 24, 8>(p, lutMode, st);
  if (Ctg % 128 == 0) return launch_conv<3, 4, 16, 8>(p, lutMode, st);
  if (Ctg % 96 == 0) return launch_conv<4, 4, 12, 8>(p, lutMode, st);
  if (Ctg % 64 == 0) return launch_conv<4, 6, 8, 8>(p, lutMode, st);
  if (Ctg > 64) return launch_conv<3, 4, 16, 8>(p, lutMode, st);
  return launch_conv<4, 6, 4, 8>(p, lutMode, st);
}

hipError_t qk_fc_aprx(const FcParams& p, int lutMode, hipStream_t st) {
  if (p.Ct % 4 || p.Cs > QCNN_MAX_CS || p.K > 256) return hipErrorInvalidValue;
  if (p.Ct >= 2048) return launch_fc<64, 4>(p, lutMode, st);
  if (p.Ct >= 256) return launch_fc<32, 4>(p, lutMode, st);
  return launch_fc<8, 4>(p, lutMode, st);
}

hipError_t qk_relu(const float* src, float* dst, size_t n, hipStream_t st) {
  const size_t n4 = n / 4;   // panel rows are 64 floats: always a multiple of 4
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
  hipLaunchKernelGGL(k_softmax, dim3(panels), dim3(64), 0, st, src, dst, panels, C);
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
