// qcnn_glue.hip — the glue kernels of the forward pass for gfx950: ReLU (src/CaffeEva.cc:1027),
// LRN (:1038), max-pool (:870), the fused LRN + max-pool of the fast path, soft-max (:1098), top-5 (:1162), the sum of
// the FC layers' split partial sums, and the conversions between the reference's NCHW / row-major host layouts and the
// 128-image panels (:1146-1160, :187-189).  All of them are streaming kernels bound by HBM (or, for LRN, by the
// expf/logf pair); the two hot kernels live in qcnn_kernels.hip.  Feature maps are panels [pixel][channel][128 images]
// (qcnn_kernels.h); a float4 lane carries four images.
#include "qcnn_kernels.h"

#include <float.h>

#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PANEL = QCNN_PANEL;              // images per panel

// dst row e = src row map[e] (the NHWC -> NCHW flatten in front of the first FC layer, src/CaffeEva.cc:187-189)
__global__ void k_permute_rows(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ map,
                               int D, int panels, int livePairs) {
  const int lane = threadIdx.x & 63;
  if (lane >= livePairs) return;
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
// in registers: every element is read exactly once.  Same summation order as k_lrn below (window j ascending;
// channels outside [0, C) contribute an exact +0.0f instead of being skipped, which leaves s > 0 bit-identical);
// the power goes through lrn_scale.
// s^(-beta) of the LRN scale.  The reference's native build evaluates exp(-beta * log(s)) in float
// (include/BlasWrapper.h:142-144), its OpenVML build calls vsPowx: the specification is the power, not one libm's way
// to it.  For beta = 0.75 — every shipped topology (src/CaffePara.cc:31,35) — the power is rsqrt(s) * sqrt(rsqrt(s)):
// v_rsq_f32 and v_sqrt_f32 are accurate to 1 ulp each, the product to <= 3.5 ulp (2e-7, the size of the error
// exp(b * log(a)) itself makes for s ~ 1..10), at 3 instructions instead of ~40.  With expf/logf the LRN kernels were
// bound by these two calls (fused LRN + pool: 0.63 ms), not by HBM.  Any other beta takes expf/logf.
template <bool B34>
__device__ __forceinline__ float lrn_scale(float s, float nbet) {
  if (B34) {
    const float r = __builtin_amdgcn_rsqf(s);
    return __fmul_rn(r, __builtin_amdgcn_sqrtf(r));
  }
  return expf(__fmul_rn(nbet, logf(s)));
}

// Native four-wide values in the LRN kernels: a float4 struct behind a conditional was scalarised into four dword loads
// and unpacked adds (36 global_load_dword per chunk of k_lrn_stream); an ext-vector loads as ONE dwordx4 and its +, * are
// v_pk_add_f32 / v_pk_mul_f32 — the same IEEE operations in the same order, half the instructions.
__device__ __forceinline__ f32x4 ld4(const float4* p, bool on) {
  f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
  if (on) v = *reinterpret_cast<const f32x4*>(p);
  return v;
}
__device__ __forceinline__ f32x4 lrn_sq(f32x4 v, float coeff) { return (v * v) * coeff; }   // (x * x) * (alpha / n), two roundings
template <bool B34>
__device__ __forceinline__ f32x4 lrn_out(f32x4 xc, f32x4 s, float nbet) {
  const f32x4 sc = {lrn_scale<B34>(s[0], nbet), lrn_scale<B34>(s[1], nbet), lrn_scale<B34>(s[2], nbet), lrn_scale<B34>(s[3], nbet)};
  return xc * sc;
}

template <int N, bool B34>
__global__ __launch_bounds__(256) void k_lrn_stream(const float4* __restrict__ src, float4* __restrict__ dst,
                                                    size_t pixels, int C, int segLen, float coeff, float nbet, float ini,
                                                    int liveQuads, int qlShift) {
  constexpr int RAD = (N - 1) / 2;
  // 1 << qlShift lanes per pixel (32 = a whole 512-byte row; a batch of a few images: just the float4 lanes it has, so
  // that a block covers 256 >> qlShift pixels instead of 8 with most lanes idle)
  const size_t px = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> qlShift;
  const int q = threadIdx.x & ((1 << qlShift) - 1);
  if (px >= pixels || q >= liveQuads) return;   // lanes of images a small batch does not have
  // blockIdx.y = channel segment [cs, ce): few pixels (a single panel of a 13x13 map) would otherwise leave most of
  // the chip idle.  segLen is a multiple of N, so channel k always lives in ring slot k % N.
  const int cs = blockIdx.y * segLen, ce = min(C, cs + segLen);
  const float4* __restrict__ x = src + px * (size_t)C * 32 + q;
  float4* __restrict__ y = dst + px * (size_t)C * 32 + q;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 raw[N], sq[N];           // ring: slot (t mod N) holds channel t
#pragma unroll
  for (int j = 0; j < N; ++j) { raw[j] = zero; sq[j] = zero; }
  // channels cs-RAD .. cs+RAD-1 enter the window before the first output of the segment
#pragma unroll
  for (int d = -RAD; d < RAD; ++d) {
    const int t = cs + d;
    const f32x4 v = ld4(x + (size_t)(t >= 0 && t < C ? t : 0) * 32, t >= 0 && t < C);
    raw[(d + N) % N] = v;
    sq[(d + N) % N] = lrn_sq(v, coeff);
  }
  for (int c0 = cs; c0 < ce; c0 += N) {
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const int c = c0 + u;                 // output channel; slot of channel k is (u + k - c) mod N
      const int tin = c + RAD;              // channel entering the window
      const f32x4 v = ld4(x + (size_t)(tin < C ? tin : 0) * 32, tin < C);
      raw[(u + RAD) % N] = v;
      sq[(u + RAD) % N] = lrn_sq(v, coeff);
      if (c < ce) {
        f32x4 sacc = {ini, ini, ini, ini};
#pragma unroll
        for (int j = 0; j < N; ++j) sacc = sacc + sq[(u - RAD + j + N) % N];   // window channel c - RAD + j, j ascending
        *reinterpret_cast<f32x4*>(y + (size_t)c * 32) = lrn_out<B34>(raw[u % N], sacc, nbet);
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
                                               int H, int W, int C, int Ho, int Wo, int knl, int stride, int pad,
                                               int liveQuads, int qlShift) {
  const size_t rows = (size_t)panels * Ho * Wo * C;
  const size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> qlShift;   // 1 << qlShift lanes per row (k_lrn_stream)
  const int q = threadIdx.x & ((1 << qlShift) - 1);
  if (r >= rows || q >= liveQuads) return;
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

// LRN followed by a 3x3 / stride 2 / pad 0 max-pool in one pass (fast path only: the normalised map is never
// materialised).  A block = a 4x4 tile of pool outputs = 9x9 source pixels, 32 images (8 float4 lanes), all channels.
// Thread (pixel, image quad) walks the channels exactly like k_lrn_stream (same arithmetic, same order) but parks N
// normalised channels at a time in an LDS slab [N][81 pixels][8 quads]; then thread (pool output, channel of the
// chunk, quad) takes the maximum of its window out of the slab (k_pool4's order and comparison) and stores it.
// Every normalised value is evaluated once per block (the one-pixel halo between tiles: 81/64 = 1.27x) instead of
// once per window that contains it (2.25x: that version lost to the two separate kernels, LABBOOK.md §3.2), and the
// map takes one HBM read instead of write + read.
#ifndef QCNN_LP_PREFETCH
#define QCNN_LP_PREFETCH 0
#endif
#ifndef QCNN_LP_PT
#define QCNN_LP_PT 4
#define QCNN_LP_Q 8
#define QCNN_LP_THREADS 704
#endif
constexpr int LP_PT = QCNN_LP_PT;               // pool outputs per tile side
constexpr int LP_IT = (LP_PT - 1) * 2 + 3;      // source pixels per tile side
constexpr int LP_PIX = LP_IT * LP_IT;
constexpr int LP_Q = QCNN_LP_Q;                 // float4 lanes (4 images each) per block (a power of two)
constexpr int LP_THREADS = QCNN_LP_THREADS;     // >= LP_PIX * LP_Q, >= LP_PT^2 * 5 * LP_Q
template <int N, bool B34>
__global__ __launch_bounds__(LP_THREADS, QCNN_LP_PREFETCH ? 2 : 1) void k_lrn_pool(const float4* __restrict__ src, float4* __restrict__ dst, int H,
                                                         int W, int C, int Ho, int Wo, int tilesX, float coeff, float nbet,
                                                         float ini, int liveQuads) {
  constexpr int RAD = (N - 1) / 2;
  __shared__ float4 slab[N * LP_PIX * LP_Q];
  const int q = threadIdx.x & (LP_Q - 1), rest = threadIdx.x / LP_Q;
  const int slices = 32 / LP_Q;
  const int slice = blockIdx.x % slices, tile = blockIdx.x / slices;
  const int panel = blockIdx.y;
  const int qg = slice * LP_Q + q;                               // float4 lane inside the 128-image row
  const bool qlive = qg < liveQuads;
  const int ty = tile / tilesX, tx = tile % tilesX;
  // LRN role: source pixel `rest` of the tile
  const int ph = ty * LP_PT * 2 + rest / LP_IT, pw = tx * LP_PT * 2 + rest % LP_IT;
  const bool lrnOn = rest < LP_PIX && ph < H && pw < W && qlive;
  const float4* __restrict__ x = src + ((size_t)panel * H * W + (size_t)(lrnOn ? ph * W + pw : 0)) * C * 32 + qg;
  // pool role: output `rest / N` of the tile, channel `rest % N` of the chunk
  const int pu = rest % N, po = rest / N;
  const int ho = ty * LP_PT + po / LP_PT, wo = tx * LP_PT + po % LP_PT;
  const bool poolOn = po < LP_PT * LP_PT && ho < Ho && wo < Wo && qlive;
  const int hU = min(H, ho * 2 + 3) - 1 - ty * LP_PT * 2, wU = min(W, wo * 2 + 3) - 1 - tx * LP_PT * 2;   // tile-relative, inclusive
  const int hL = (po / LP_PT) * 2, wL = (po % LP_PT) * 2;
  float4* __restrict__ y = dst + ((size_t)panel * Ho * Wo + (size_t)(poolOn ? ho * Wo + wo : 0)) * C * 32 + qg;

  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 raw[N], sq[N];           // ring: slot (t mod N) holds channel t (k_lrn_stream)
#pragma unroll
  for (int j = 0; j < N; ++j) { raw[j] = zero; sq[j] = zero; }
#pragma unroll
  for (int d = -RAD; d < RAD; ++d) {
    const bool on = lrnOn && d >= 0 && d < C;
    const f32x4 v = ld4(x + (size_t)(on ? d : 0) * 32, on);
    raw[(d + N) % N] = v;
    sq[(d + N) % N] = lrn_sq(v, coeff);
  }
  f32x4* __restrict__ slab4 = reinterpret_cast<f32x4*>(slab);
#if QCNN_LP_PREFETCH
  f32x4 nxt[N];                                    // the N channels entering the window in the NEXT chunk: in flight under
#pragma unroll                                     // this chunk's arithmetic, its pool phase and both barriers
  for (int u = 0; u < N; ++u) nxt[u] = ld4(x + (size_t)(u + RAD < C ? u + RAD : 0) * 32, lrnOn && u + RAD < C);
#endif
  for (int c0 = 0; c0 < C; c0 += N) {
    if (lrnOn) {
#pragma unroll
      for (int u = 0; u < N; ++u) {
        const int c = c0 + u;
        const int tin = c + RAD;
#if QCNN_LP_PREFETCH
        const f32x4 v = nxt[u];
        {
          const int tn = c0 + N + u + RAD;          // the same slot of the next chunk
          nxt[u] = ld4(x + (size_t)(tn < C ? tn : 0) * 32, tn < C);
        }
#else
        const f32x4 v = ld4(x + (size_t)(tin < C ? tin : 0) * 32, tin < C);
#endif
        raw[(u + RAD) % N] = v;
        sq[(u + RAD) % N] = lrn_sq(v, coeff);
        if (c < C) {
          f32x4 sacc = {ini, ini, ini, ini};
#pragma unroll
          for (int j = 0; j < N; ++j) sacc = sacc + sq[(u - RAD + j + N) % N];
          slab4[(u * LP_PIX + rest) * LP_Q + q] = lrn_out<B34>(raw[u % N], sacc, nbet);
        }
      }
    }
    __syncthreads();
    if (poolOn && c0 + pu < C) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      bool first = true;
      for (int h = hL; h <= hU; ++h)
        for (int w = wL; w <= wU; ++w) {
          const float4 sv = slab[(pu * LP_PIX + h * LP_IT + w) * LP_Q + q];
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
      y[(size_t)(c0 + pu) * 32] = v;
    }
    __syncthreads();
  }
}

// Softmax through LDS: a block = 32 images x 32 class lanes.  expf of every logit in parallel into an LDS tile
// [C][33] (odd row stride: both the image-major and the class-major access below are conflict-free), the reference's
// SEQUENTIAL float sum over the classes (src/CaffeEva.cc:1107-1114) by one thread per image out of LDS — reads
// issued sixteen ahead of the dependent adds —, then the division in parallel.  Same values as k_softmax.
constexpr int SM_LD = 33;
__global__ __launch_bounds__(1024) void k_softmax_lds(const float* __restrict__ src, float* __restrict__ dst, int C) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* tile = reinterpret_cast<float*>(lds);            // [C][33]
  float* sums = tile + (size_t)C * SM_LD;                 // [32]
  const int img = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const size_t off = (size_t)blockIdx.y * C * PANEL + blockIdx.x * 32 + img;
  const float* x = src + off;
  float* y = dst + off;
  int c = cl;
  for (; c + 96 < C; c += 128) {                          // four independent loads in flight per thread
    const float v0 = x[(size_t)c * PANEL], v1 = x[(size_t)(c + 32) * PANEL], v2 = x[(size_t)(c + 64) * PANEL],
                v3 = x[(size_t)(c + 96) * PANEL];
    tile[c * SM_LD + img] = expf(v0);
    tile[(c + 32) * SM_LD + img] = expf(v1);
    tile[(c + 64) * SM_LD + img] = expf(v2);
    tile[(c + 96) * SM_LD + img] = expf(v3);
  }
  for (; c < C; c += 32) tile[c * SM_LD + img] = expf(x[(size_t)c * PANEL]);
  __syncthreads();
  if (cl == 0) {
    float sum = 0.0f;
    int k = 0;
    for (; k + 16 <= C; k += 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = tile[(k + j) * SM_LD + img];
#pragma unroll
      for (int j = 0; j < 16; ++j) sum = __fadd_rn(sum, v[j]);
    }
    for (; k < C; ++k) sum = __fadd_rn(sum, tile[k * SM_LD + img]);
    sums[img] = sum;
  }
  __syncthreads();
  const float sum = sums[img];
  for (c = cl; c < C; c += 32) y[(size_t)c * PANEL] = __fdiv_rn(tile[c * SM_LD + img], sum);
}

// Top-5 through LDS: a block stages 32 images x C classes as [C][33]; then every half-wave owns ONE image, its 32
// lanes sweep the classes c = lane, lane + 32, ... for their first maximum (strict '<' from FLT_MIN), a butterfly
// over the 32 lanes merges the candidates (larger value, then lower index = the sequential sweep's first
// occurrence), the winner is zeroed (src/CaffeEva.cc:1173-1188).  An image's column is touched by its own
// half-wave only: no workgroup barrier inside the five sweeps.
__global__ __launch_bounds__(1024) void k_top5_lds(const float* __restrict__ prob, uint16_t* __restrict__ out, int n,
                                                   int C) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* tile = reinterpret_cast<float*>(lds);            // [C][33]
  {
    const int img = threadIdx.x & 31, cl = threadIdx.x >> 5;
    const float* x = prob + (size_t)blockIdx.y * C * PANEL + blockIdx.x * 32 + img;
    int c = cl;
    for (; c + 96 < C; c += 128) {
      const float v0 = x[(size_t)c * PANEL], v1 = x[(size_t)(c + 32) * PANEL], v2 = x[(size_t)(c + 64) * PANEL],
                  v3 = x[(size_t)(c + 96) * PANEL];
      tile[c * SM_LD + img] = v0;
      tile[(c + 32) * SM_LD + img] = v1;
      tile[(c + 64) * SM_LD + img] = v2;
      tile[(c + 96) * SM_LD + img] = v3;
    }
    for (; c < C; c += 32) tile[c * SM_LD + img] = x[(size_t)c * PANEL];
  }
  __syncthreads();
  const int img = threadIdx.x >> 5, cl = threadIdx.x & 31;
  const int gi = blockIdx.y * PANEL + blockIdx.x * 32 + img;
  for (int r = 0; r < 5; ++r) {
    float best = FLT_MIN;
    int bi = 0;
    for (int c = cl; c < C; c += 32) {
      const float v = tile[c * SM_LD + img];
      if (best < v) { best = v; bi = c; }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
      const float v = __shfl_xor(best, d, 32);
      const int vi = __shfl_xor(bi, d, 32);
      if (best < v || (best == v && vi < bi)) { best = v; bi = vi; }
    }
    if ((bi & 31) == cl) {                                // the lane that sweeps class bi: it reads the zero next round
      tile[bi * SM_LD + img] = 0.0f;
      if (gi < n) out[(size_t)gi * 5 + r] = (uint16_t)bi;
    }
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
  const int e = e0 + lane;
#pragma unroll
  for (int b = 0; b < PANEL / 4; b += 8) {          // eight independent loads in flight per thread
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int img = panel * PANEL + wave + 4 * (b + u);
      v[u] = (img < n && e < E) ? in[(size_t)img * E + e] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) tile[wave + 4 * (b + u)][lane] = v[u];
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
#pragma unroll
  for (int b = 0; b < PANEL / 4; b += 8) {          // eight independent loads in flight per thread
    uint8_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int img = panel * PANEL + wave + 4 * (b + u);
      v[u] = (img < n && e < E) ? in[(size_t)img * srcImg + soff] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int img = panel * PANEL + wave + 4 * (b + u);
      tile[wave + 4 * (b + u)][lane] = (img < n && e < E) ? ((float)v[u] - m) : 0.0f;
    }
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

}  // namespace

hipError_t qk_sum_partials(const float* partial, float* dst, int msplit, size_t n, int relu, hipStream_t st) {
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_sum_partials, dim3(blocks ? blocks : 1), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(dst), msplit, n4, relu);
  return hipGetLastError();
}

hipError_t qk_permute_rows(const float* src, float* dst, const int* map, int D, int panels, int live, hipStream_t st) {
  const size_t rows = (size_t)panels * D;
  const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_permute_rows, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, map, D, panels, (live + 1) / 2);
  return hipGetLastError();
}

hipError_t qk_relu(const float* src, float* dst, size_t n, hipStream_t st) {
  const size_t n4 = n / 4;   // panel rows are 128 floats: always a multiple of 4
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_relu, dim3(blocks ? blocks : 1), dim3(256), 0, st, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n4);
  return hipGetLastError();
}

namespace {
// log2 of the float4 lanes a row gets in the streaming glue kernels: 32 (the whole 512-byte row) unless the batch is a single
// panel with fewer live images
int live_shift(int live) {
  const int quads = (live + 3) / 4;
  int s = 0;
  while ((1 << s) < quads) ++s;
  return s > 5 ? 5 : s;
}
}  // namespace

hipError_t qk_lrn(const float* src, float* dst, int panels, int HW, int C, int lrnSiz, float alp, float bet, float ini,
                  int live, hipStream_t st) {
  const size_t rows = (size_t)panels * HW * C;
  const float coeff = alp / lrnSiz;   // float / int, as src/CaffeEva.cc:1055
  if (lrnSiz == 5 || lrnSiz == 3) {   // streaming kernel: 8 pixels per block (more when a small batch has few float4 lanes)
    const size_t pixels = (size_t)panels * HW;
    const int qlShift = live_shift(live);
    const size_t blocks = ((pixels << qlShift) + 255) / 256;
    // channel segments (blockIdx.y) until ~8 blocks per CU exist; a segment keeps >= 4 window lengths of channels (a few
    // images: >= 2 — the walk along the channels is a chain of dependent loads, short segments cut it)
    int segs = (int)std::min<size_t>((2048 + blocks - 1) / blocks, (size_t)std::max(1, C / ((qlShift < 5 ? 2 : 4) * lrnSiz)));
    int segLen = ((C + segs - 1) / segs + lrnSiz - 1) / lrnSiz * lrnSiz;
    segs = (C + segLen - 1) / segLen;
    const dim3 grid((unsigned)blocks, (unsigned)segs);
    auto kern = k_lrn_stream<5, false>;
    if (lrnSiz == 5) kern = (bet == 0.75f) ? k_lrn_stream<5, true> : k_lrn_stream<5, false>;
    else kern = (bet == 0.75f) ? k_lrn_stream<3, true> : k_lrn_stream<3, false>;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), pixels, C, segLen, coeff, -bet, ini, (live + 3) / 4, qlShift);
    return hipGetLastError();
  }
  const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_lrn, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, rows, C, lrnSiz, coeff, -bet, ini);
  return hipGetLastError();
}

int qk_lrn_pool_blocks(int Ho, int Wo) { return ((Ho + LP_PT - 1) / LP_PT) * ((Wo + LP_PT - 1) / LP_PT) * (32 / LP_Q); }

// LRN + the 3x3 / stride 2 / pad 0 max-pool behind it; the caller checked the window (qk_lrn_pool_blocks per panel)
hipError_t qk_lrn_pool(const float* src, float* dst, int panels, int H, int W, int C, int Ho, int Wo, int lrnSiz, float alp,
                       float bet, float ini, int live, hipStream_t st) {
  if (lrnSiz != 5 && lrnSiz != 3) return hipErrorInvalidValue;
  const float coeff = alp / lrnSiz;   // float / int, as src/CaffeEva.cc:1055
  const int tilesX = (Wo + LP_PT - 1) / LP_PT;
  const dim3 grid((unsigned)qk_lrn_pool_blocks(Ho, Wo), (unsigned)panels);
  auto kern = k_lrn_pool<5, false>;
  if (lrnSiz == 5) kern = (bet == 0.75f) ? k_lrn_pool<5, true> : k_lrn_pool<5, false>;
  else kern = (bet == 0.75f) ? k_lrn_pool<3, true> : k_lrn_pool<3, false>;
  hipLaunchKernelGGL(kern, grid, dim3(LP_THREADS), 0, st, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), H, W, C, Ho, Wo, tilesX, coeff, -bet, ini, (live + 3) / 4);
  return hipGetLastError();
}

hipError_t qk_pool(const float* src, float* dst, int panels, int H, int W, int C, int Ho, int Wo, int knl, int stride,
                   int pad, int live, hipStream_t st) {
  const size_t rows = (size_t)panels * Ho * Wo * C;
  if ((rows + 7) / 8 < (size_t)1 << 31) {
    const int qlShift = live_shift(live);
    hipLaunchKernelGGL(k_pool4, dim3((unsigned)(((rows << qlShift) + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), panels, H, W, C, Ho, Wo, knl,
                       stride, pad, (live + 3) / 4, qlShift);
    return hipGetLastError();
  }
  const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
  hipLaunchKernelGGL(k_pool, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, panels, H, W, C, Ho, Wo, knl,
                     stride, pad);
  return hipGetLastError();
}

hipError_t qk_softmax(const float* src, float* dst, int panels, int C, int live, hipStream_t st) {
  const size_t shm = ((size_t)C * SM_LD + 32) * sizeof(float);
  if (shm <= 160 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_softmax_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_softmax_lds, dim3((live + 31) / 32, panels), dim3(1024), shm, st, src, dst, C);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_softmax, dim3((panels * PANEL + 63) / 64), dim3(64), 0, st, src, dst, panels, C);
  return hipGetLastError();
}

hipError_t qk_top5(const float* prob, uint16_t* out, int n, int C, hipStream_t st) {
  const size_t shm = (size_t)C * SM_LD * sizeof(float);
  if (shm <= 160 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_top5_lds),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_top5_lds, dim3(n <= PANEL ? (n + 31) / 32 : PANEL / 32, panels_of(n)), dim3(1024), shm, st, prob, out, n, C);
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
