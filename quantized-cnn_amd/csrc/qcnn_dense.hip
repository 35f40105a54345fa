// qcnn_dense.hip — the reference's PRECISE path (CaffeEva::Init(false)) on the same panel layout: the exact conv / FC
// layers the approximate pass is compared against on the device itself.
//
//   k_dense   CalcFeatMap_ConvPrec  src/CaffeEva.cc:681-758  (im2col :1195-1243 + cblas_sgemm_nn + bias)
//             CalcFeatMap_FCntPrec  src/CaffeEva.cc:932-966  (cblas_sgemm_nt + bias)   — an FC layer is a 1x1 conv on a 1x1 map
//
// No im2col buffer: a workgroup owns one output position, 64 output channels of one group and one 128-image panel; a
// wave owns 16 channels x 128 images = eight v_mfma_f32_16x16x4_f32 accumulator tiles and walks the taps and the input
// channels four at a time — A = weights [16 channels x 4 inputs] (uploaded as [group][tap][input][channel], so that a
// wave's operand is two 64-byte segments), B = activations [4 inputs x 16 images], read straight from the panel rows.
// Sums are fused multiply-adds in (tap, input) order instead of the reference's separately rounded (input, tap) order:
// equal to the reference within the north-star tolerance (1e-4), like every MFMA path of this library.
// The reference's im2col drops some taps at output row / column 0 of strided layers (truncating division on a negative
// numerator, :1219,1223); that is reproduced: the reference's behaviour is the specification.
#include "qcnn_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PANEL = QCNN_PANEL;

__global__ __launch_bounds__(256) void k_dense(DenseParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int pos = blockIdx.x, panel = blockIdx.z;
  const int Cg = p.Cin / p.grp, Ctg = p.Ct / p.grp;
  const int chunks = (Ctg + 63) / 64;
  const int grp = blockIdx.y / chunks, chunk = blockIdx.y % chunks;
  const int c0 = chunk * 64 + wave * 16;                       // first channel of this wave inside the group
  if (c0 >= Ctg) return;
  const int ho = pos / p.Wo, wo = pos % p.Wo;
  const float* __restrict__ src = p.src + (size_t)panel * p.H * p.W * p.Cin * PANEL;
  const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 acc[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) acc[it] = zero;
  const bool chOk = c0 + li < Ctg;
  for (int kh = 0; kh < p.knl; ++kh) {
    const int hi = ho * p.stride - p.pad + kh;
    // rows this tap fills in the reference's im2col buffer (src/CaffeEva.cc:1219-1222, C integer division as written)
    const int hoL = max(0, (p.pad - kh - 1) / p.stride + 1), hoU = min(p.Ho - 1, (p.pad - kh + p.H - 1) / p.stride);
    if (ho < hoL || ho > hoU) continue;
    for (int kw = 0; kw < p.knl; ++kw) {
      const int wi = wo * p.stride - p.pad + kw;
      const int woL = max(0, (p.pad - kw - 1) / p.stride + 1), woU = min(p.Wo - 1, (p.pad - kw + p.W - 1) / p.stride);
      if (wo < woL || wo > woU) continue;
      const float* __restrict__ xr = src + ((size_t)(hi * p.W + wi) * p.Cin + grp * Cg) * PANEL;
      const float* __restrict__ wr = p.wt + ((size_t)(grp * p.knl * p.knl + kh * p.knl + kw) * Cg) * Ctg + c0;
      for (int ci = 0; ci < Cg; ci += 4) {
        const bool kOk = ci + lk < Cg;
        const float a = (kOk && chOk) ? wr[(size_t)(ci + lk) * Ctg + li] : 0.0f;
        const float* __restrict__ xk = xr + (size_t)(kOk ? ci + lk : 0) * PANEL + li;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const float b = kOk ? xk[it * 16] : 0.0f;
          acc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[it], 0, 0, 0);
        }
      }
    }
  }
  // D[(lane >> 4) * 4 + r][lane & 15]: channel c0 + 4 * lk + r, image it * 16 + li
  float* __restrict__ dst = p.dst + ((size_t)panel * p.Ho * p.Wo + pos) * p.Ct * PANEL;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ch = c0 + 4 * lk + r;
    if (ch < Ctg) {
      const float bv = p.bias[grp * Ctg + ch];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        float v = acc[it][r] + bv;
        if (p.relu) v = (0.0f < v) ? v : 0.0f;
        dst[(size_t)(grp * Ctg + ch) * PANEL + it * 16 + li] = v;
      }
    }
  }
}

}  // namespace

hipError_t qk_dense(const DenseParams& p, hipStream_t st) {
  if (p.grp <= 0 || p.Cin % p.grp || p.Ct % p.grp) return hipErrorInvalidValue;
  const int Ctg = p.Ct / p.grp;
  const dim3 grid((unsigned)(p.Ho * p.Wo), (unsigned)(((Ctg + 63) / 64) * p.grp), (unsigned)p.panels);
  hipLaunchKernelGGL(k_dense, grid, dim3(256), 0, st, p);
  return hipGetLastError();
}
