/* TEST INFRASTRUCTURE — see qcnn_oracle.h.  Plain C99 restatement of the reference's approximate
 * forward pass; every routine cites the reference file:line it follows and keeps the reference's
 * floating-point operation ORDER (separate multiply / add, sequential sums), so that on the same
 * host it reproduces the compiled reference bit for bit (build with -ffp-contract=off).
 */
#include "qcnn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

int qo_conv_out(int in, int knl, int stride, int pad) { return (in + 2 * pad - knl) / stride + 1; }

int qo_pool_out(int in, int knl, int stride, int pad) {
  return (int)ceil((in + 2 * pad - knl) / (double)stride) + 1;
}

/* ---- tolerance study of BASELINE.json configs[4] (fp16 LUT storage / accumulation) -------------------------
 * NOT part of the restatement: with both switches at 0 (the default, and the only setting the parity tests
 * use) every function below is the reference's arithmetic.  qo_study_mode(1, 0) rounds every table entry to
 * fp16 when it is stored (what a half-size LDS table would hold); qo_study_mode(1, 1) additionally keeps the
 * running sums of the look-up loops in fp16 (rounded after every addition; the bias start value too). */
static int g_lutF16 = 0, g_accF16 = 0;
void qo_study_mode(int lutF16, int accF16) { g_lutF16 = lutF16; g_accF16 = accF16; }
/* round-to-nearest-even to IEEE binary16 and back (gcc 11 has no _Float16 on x86-64): 10 mantissa bits for normal
 * halves (|v| >= 2^-14), a fixed quantum of 2^-24 below that, infinity beyond 65504 + half a step */
static float rnd16(float v) {
  if (v != v) return v;
  const float a = fabsf(v);
  if (a >= 65520.0f) return v < 0 ? -INFINITY : INFINITY;
  if (a < 6.103515625e-05f) {                       /* subnormal half range: multiples of 2^-24 */
    const float q = 5.9604644775390625e-08f;
    return nearbyintf(v / q) * q;                  /* default rounding mode = nearest even */
  }
  uint32_t u;
  memcpy(&u, &v, 4);
  u += 0x00000fffu + ((u >> 13) & 1u);             /* keep 10 of the 23 mantissa bits, ties to even */
  u &= 0xffffe000u;
  float r;
  memcpy(&r, &u, 4);
  return r;
}

/* src/CaffeEva.cc:1261-1296.  Table entry = ((0 + x0*c0) + x1*c1) + ... over the CsEff dims that
 * exist (:1277); saxpy is y += a*x with a rounded product (include/BlasWrapper.h:164-184). */
void qo_lut_build(const float* data, int P, int D, const float* ctrd, int M, int Cs, int K, float* lut) {
  for (int m = 0; m < M; ++m) {
    const int d0 = Cs * m;
    const int dsel = imin(D - d0, Cs);
    for (int p = 0; p < P; ++p) {
      const float* x = data + (size_t)p * D + d0;
      float* y = lut + ((size_t)p * M + m) * K;
      const float* c = ctrd + (size_t)m * Cs * K;
      for (int k = 0; k < K; ++k) y[k] = 0.0f;
      for (int d = 0; d < dsel; ++d) {
        const float a = x[d];
        for (int k = 0; k < K; ++k) {
          const float prod = a * c[(size_t)d * K + k];
          y[k] = y[k] + prod;
        }
      }
      if (g_lutF16)
        for (int k = 0; k < K; ++k) y[k] = rnd16(y[k]);
    }
  }
}

/* src/CaffeEva.cc:760-868 */
void qo_conv_aprx(const float* src, int B, int H, int W, int Cin, int knl, int stride, int pad, int grp,
                  int Ct, const float* bias, const float* ctrd, int M, int Cs, int K,
                  const uint8_t* asmt, float* dst, float* lutScratch) {
  const int Ho = qo_conv_out(H, knl, stride, pad), Wo = qo_conv_out(W, knl, stride, pad);
  const int Cg = Cin / grp, Ctg = Ct / grp;
  const size_t P = (size_t)B * H * W;
  float* slice = (float*)malloc(sizeof(float) * P * Cg);
  for (int g = 0; g < grp; ++g) {
    /* channel slice of this group (:802-809), the SAME codebook for every group (:787,810) */
    for (size_t p = 0; p < P; ++p) memcpy(slice + p * Cg, src + p * Cin + (size_t)g * Cg, sizeof(float) * Cg);
    qo_lut_build(slice, (int)P, Cg, ctrd, M, Cs, K, lutScratch);
    const int c0 = g * Ctg;
    for (int ho = 0; ho < Ho; ++ho) {
      for (int wo = 0; wo < Wo; ++wo) {
        const int hs = ho * stride - pad, ws = wo * stride - pad;
        const int khL = imax(0, -hs), khU = imin(knl - 1, H - 1 - hs);   /* :824-827 */
        const int kwL = imax(0, -ws), kwU = imin(knl - 1, W - 1 - ws);
        for (int b = 0; b < B; ++b) {
          float* o = dst + (((size_t)b * Ho + ho) * Wo + wo) * Ct + c0;
          memcpy(o, bias + c0, sizeof(float) * Ctg);                     /* :834 */
          if (g_accF16)
            for (int c = 0; c < Ctg; ++c) o[c] = rnd16(o[c]);
          for (int kh = khL; kh <= khU; ++kh) {
            for (int kw = kwL; kw <= kwU; ++kw) {
              const float* t = lutScratch + (((size_t)b * H + (hs + kh)) * W + (ws + kw)) * M * K;
              const uint8_t* a = asmt + ((size_t)(kh * knl + kw) * M) * Ct + c0;
              for (int m = 0; m < M; ++m) {
                if (g_accF16)
                  for (int c = 0; c < Ctg; ++c) o[c] = rnd16(o[c] + t[a[c]]);
                else
                  for (int c = 0; c < Ctg; ++c) o[c] = o[c] + t[a[c]];    /* :849-858 */
                t += K;
                a += Ct;
              }
            }
          }
        }
      }
    }
  }
  free(slice);
}

/* src/CaffeEva.cc:968-1025 */
void qo_fc_aprx(const float* src, int B, int D, int Ct, const float* bias, const float* ctrd,
                int M, int Cs, int K, const uint8_t* asmt, float* dst, float* lutScratch) {
  qo_lut_build(src, B, D, ctrd, M, Cs, K, lutScratch);
  for (int b = 0; b < B; ++b) {
    float* o = dst + (size_t)b * Ct;
    memcpy(o, bias, sizeof(float) * Ct);
    const float* t = lutScratch + (size_t)b * M * K;
    const uint8_t* a = asmt;
    if (g_accF16)
      for (int c = 0; c < Ct; ++c) o[c] = rnd16(o[c]);
    for (int m = 0; m < M; ++m) {
      if (g_accF16)
        for (int c = 0; c < Ct; ++c) o[c] = rnd16(o[c] + t[a[c]]);
      else
        for (int c = 0; c < Ct; ++c) o[c] = o[c] + t[a[c]];
      t += K;
      a += Ct;
    }
  }
}

/* src/CaffeEva.cc:1027-1036: std::max(0.0f, x) == (0 < x) ? x : 0 */
void qo_relu(const float* src, int n, float* dst) {
  for (int i = 0; i < n; ++i) dst[i] = (0.0f < src[i]) ? src[i] : 0.0f;
}

/* src/CaffeEva.cc:1038-1089 with the native helpers of include/BlasWrapper.h:101-162:
 * sq = x*x; sq *= alpha/n; s = k; s += sq[c+j] for j ascending; s = expf(-beta * logf(s)); y = x*s */
void qo_lrn(const float* src, int B, int H, int W, int C, int lrnSiz, float alp, float bet, float ini,
            float* dst) {
  const int rad = (lrnSiz - 1) / 2;
  const int cext = C + 2 * rad;
  float* ext = (float*)calloc((size_t)cext, sizeof(float));
  float* sum = (float*)malloc(sizeof(float) * C);
  const float coeff = alp / lrnSiz;
  const float nbet = -bet;
  const size_t P = (size_t)B * H * W;
  for (size_t p = 0; p < P; ++p) {
    const float* x = src + p * C;
    float* y = dst + p * C;
    for (int c = 0; c < C; ++c) ext[rad + c] = x[c] * x[c];
    for (int c = 0; c < C; ++c) ext[rad + c] *= coeff;
    for (int c = 0; c < C; ++c) sum[c] = ini;
    for (int j = 0; j < lrnSiz; ++j)
      for (int c = 0; c < C; ++c) sum[c] = sum[c] + ext[c + j];
    for (int c = 0; c < C; ++c) sum[c] = expf(nbet * logf(sum[c]));
    for (int c = 0; c < C; ++c) y[c] = x[c] * sum[c];
  }
  free(ext);
  free(sum);
}

/* src/CaffeEva.cc:870-921: ceil-mode output size, window clipped to the image */
void qo_pool(const float* src, int B, int H, int W, int C, int knl, int stride, int pad, float* dst) {
  const int Ho = qo_pool_out(H, knl, stride, pad), Wo = qo_pool_out(W, knl, stride, pad);
  for (int ho = 0; ho < Ho; ++ho) {
    const int hL = imax(0, ho * stride - pad), hU = imin(H, ho * stride + knl - pad) - 1;
    for (int wo = 0; wo < Wo; ++wo) {
      const int wL = imax(0, wo * stride - pad), wU = imin(W, wo * stride + knl - pad) - 1;
      for (int b = 0; b < B; ++b) {
        float* o = dst + (((size_t)b * Ho + ho) * Wo + wo) * C;
        int first = 1;
        for (int h = hL; h <= hU; ++h) {
          for (int w = wL; w <= wU; ++w) {
            const float* s = src + (((size_t)b * H + h) * W + w) * C;
            if (first) {
              memcpy(o, s, sizeof(float) * C);
              first = 0;
            } else {
              for (int c = 0; c < C; ++c) o[c] = (s[c] < o[c]) ? o[c] : s[c];   /* std::max(s, o) */
            }
          }
        }
      }
    }
  }
}

/* src/CaffeEva.cc:1098-1116: no max-shift; float running sum */
void qo_softmax(const float* src, int B, int C, float* dst) {
  for (int b = 0; b < B; ++b) {
    const float* x = src + (size_t)b * C;
    float* y = dst + (size_t)b * C;
    float sum = 0.0f;
    for (int c = 0; c < C; ++c) {
      y[c] = expf(x[c]);
      sum += y[c];
    }
    for (int c = 0; c < C; ++c) y[c] /= sum;
  }
}

/* src/CaffeEva.cc:1173-1188: five linear arg-max sweeps, strict '<' from FLT_MIN, lowest index wins */
void qo_top5(const float* prob, int C, uint16_t* out5) {
  float* p = (float*)malloc(sizeof(float) * C);
  memcpy(p, prob, sizeof(float) * C);
  for (int r = 0; r < 5; ++r) {
    float best = FLT_MIN;
    uint16_t bi = 0;
    for (int c = 0; c < C; ++c) {
      if (best < p[c]) {
        best = p[c];
        bi = (uint16_t)c;
      }
    }
    p[bi] = 0.0f;
    out5[r] = bi;
  }
  free(p);
}

/* src/CaffeEva.cc:556-557: [M][K][Cs] -> Permute(0,2,1) -> [M][Cs][K] */
void qo_prep_ctrd(const float* f, int M, int K, int Cs, float* o) {
  for (int m = 0; m < M; ++m)
    for (int d = 0; d < Cs; ++d)
      for (int k = 0; k < K; ++k) o[((size_t)m * Cs + d) * K + k] = f[((size_t)m * K + k) * Cs + d];
}

/* src/CaffeEva.cc:585-586: [Ct][kh][kw][M] -> Permute(1,2,3,0) -> [kh][kw][M][Ct] */
void qo_prep_asmt_conv(const uint8_t* f, int Ct, int kh, int kw, int M, uint8_t* o) {
  for (int c = 0; c < Ct; ++c)
    for (int y = 0; y < kh; ++y)
      for (int x = 0; x < kw; ++x)
        for (int m = 0; m < M; ++m)
          o[(((size_t)y * kw + x) * M + m) * Ct + c] = f[(((size_t)c * kh + y) * kw + x) * M + m];
}

/* src/CaffeEva.cc:610-611: [Ct][M] -> Permute(1,0) -> [M][Ct] */
void qo_prep_asmt_fc(const uint8_t* f, int Ct, int M, uint8_t* o) {
  for (int c = 0; c < Ct; ++c)
    for (int m = 0; m < M; ++m) o[(size_t)m * Ct + c] = f[(size_t)c * M + m];
}

/* include/FileIO.h:128-166, restated as a bit cursor: 4096-byte blocks, floor(32768/bits) values per
 * block, MSB first, no value straddles a block.  Output is the stored (0-based) value, i.e. what is
 * left after the reader's +1 (:165) and LoadLayerPara's -1 (src/CaffePara.cc:285-288). */
void qo_cbn_decode(const uint8_t* blocks, int n, int bits, uint8_t* out) {
  const int per = 4096 * 8 / bits;
  for (int i = 0; i < n; ++i) {
    const int blk = i / per, j = i % per;
    const uint8_t* b = blocks + (size_t)blk * 4096;
    const int bit0 = j * bits;
    unsigned v = 0;
    for (int t = 0; t < bits; ++t) {
      const int bit = bit0 + t;
      v = (v << 1) | ((b[bit >> 3] >> (7 - (bit & 7))) & 1u);
    }
    out[i] = (uint8_t)v;
  }
}

/* ------------------------------------------------------------------ precise path ------------ */

/* CaffeEva::CalcFeatMap_ConvPrec, src/CaffeEva.cc:681-758: per image and group, im2col (CvtFeatMapToFeatBuf
 * :1195-1243, zero fill outside the map and — a quirk of its index arithmetic — at output row / column 0 for some taps of
 * strided layers, see below; buffer row = (channel, kh, kw)) then the native cblas_sgemm_nn
 * (src/BlasWrapper.cc:55-74: C = 0, then for every k in ascending order C += (A[k] * alpha) * B[k]), then the bias
 * (:737-745).  src [B][H][W][Cin] NHWC, knl [Ct][Cin/grp][kh][kw] (file layout of convKnl.NN.bin), dst [B][Ho][Wo][Ct]. */
void qo_conv_prec(const float* src, int B, int H, int W, int Cin, int knl, int stride, int pad, int grp, int Ct,
                  const float* bias, const float* kn, float* dst) {
  const int Ho = qo_conv_out(H, knl, stride, pad), Wo = qo_conv_out(W, knl, stride, pad);
  const int Cg = Cin / grp, Ctg = Ct / grp, Kd = Cg * knl * knl;
  for (int b = 0; b < B; ++b)
    for (int g = 0; g < grp; ++g)
      for (int c = 0; c < Ctg; ++c) {
        const float* wr = kn + (size_t)(g * Ctg + c) * Kd;
        for (int ho = 0; ho < Ho; ++ho)
          for (int wo = 0; wo < Wo; ++wo) {
            float acc = 0.0f;                                   /* pc[in] *= beta with beta = 0 */
            for (int k = 0; k < Kd; ++k) {                      /* ik ascending: (channel, kh, kw) */
              const int kw = k % knl, kh = k / knl % knl, ci = k / knl / knl;
              const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
              /* Rows / columns of the im2col buffer a tap fills (:1219-1226).  The lower bounds use C's truncating
               * division on a NEGATIVE numerator: for 0 <= kh - pad <= stride - 2 they come out as 1 instead of 0, so
               * with a stride >= 2 those taps never reach output row / column 0 (the buffer keeps its zero there).
               * Reproduced as is: the reference's behaviour is the specification of this path. */
              const int hoL = imax(0, (pad - kh - 1) / stride + 1), hoU = imin(Ho - 1, (pad - kh + H - 1) / stride);
              const int woL = imax(0, (pad - kw - 1) / stride + 1), woU = imin(Wo - 1, (pad - kw + W - 1) / stride);
              const float x = (ho >= hoL && ho <= hoU && wo >= woL && wo <= woU)
                                  ? src[(((size_t)b * H + hi) * W + wi) * Cin + g * Cg + ci] : 0.0f;
              const float va = wr[k] * 1.0f;                    /* pa[ik] * alpha */
              acc = acc + va * x;
            }
            dst[(((size_t)b * Ho + ho) * Wo + wo) * Ct + g * Ctg + c] = acc + bias[g * Ctg + c];
          }
      }
}

/* CaffeEva::CalcFeatMap_FCntPrec, src/CaffeEva.cc:932-966: cblas_sgemm_nt (src/BlasWrapper.cc:77-97: val = sum over k
 * ascending of A[k] * B[k], C = C * 0 + val * alpha), then the bias.  src [B][D], wei [Ct][D] (fcntWei.NN.bin), dst [B][Ct] */
void qo_fc_prec(const float* src, int B, int D, int Ct, const float* bias, const float* wei, float* dst) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < Ct; ++c) {
      float val = 0.0f;
      for (int k = 0; k < D; ++k) val = val + src[(size_t)b * D + k] * wei[(size_t)c * D + k];
      dst[(size_t)b * Ct + c] = (0.0f + val * 1.0f) + bias[c];
    }
}

/* ------------------------------------------------------------------ network runner ---------- */

typedef struct {
  float* bias;
  float* ctrd;     /* [M][Cs][K] */
  uint8_t* asmt;   /* permuted */
  int M, K, Cs;
  float* dense;    /* precise path: conv kernels [Ct][Cg][kh][kw] / FC weights [Ct][D] in file layout, or NULL */
} QoParam;

typedef struct {
  int inC, inH, inW, L;
  QoLayer* ly;
  QoParam* pa;
  int (*dims)[3];  /* H, W, C of fm[0..L] */
  float** fm;      /* NHWC, sized for curB */
  int curB;
  int firstFc;
} QoNet;

void* qo_net_create(int inC, int inH, int inW, int L, const QoLayer* layers) {
  QoNet* n = (QoNet*)calloc(1, sizeof(QoNet));
  n->inC = inC; n->inH = inH; n->inW = inW; n->L = L;
  n->ly = (QoLayer*)malloc(sizeof(QoLayer) * L);
  memcpy(n->ly, layers, sizeof(QoLayer) * L);
  n->pa = (QoParam*)calloc((size_t)L, sizeof(QoParam));
  n->dims = (int (*)[3])malloc(sizeof(int[3]) * (L + 1));
  n->fm = (float**)calloc((size_t)L + 1, sizeof(float*));
  n->firstFc = -1;
  int h = inH, w = inW, c = inC;
  n->dims[0][0] = h; n->dims[0][1] = w; n->dims[0][2] = c;
  for (int l = 0; l < L; ++l) {                                   /* src/CaffeEva.cc:357-391 */
    const QoLayer* y = &n->ly[l];
    if (y->type == QO_CONV) {
      h = qo_conv_out(h, y->knlSiz, y->stride, y->padSiz);
      w = qo_conv_out(w, y->knlSiz, y->stride, y->padSiz);
      c = y->knlCnt;
    } else if (y->type == QO_POOL) {
      h = qo_pool_out(h, y->knlSiz, y->stride, y->padSiz);
      w = qo_pool_out(w, y->knlSiz, y->stride, y->padSiz);
    } else if (y->type == QO_FCNT) {
      if (n->firstFc < 0) n->firstFc = l;
      h = 1; w = 1; c = y->nodCnt;
    }
    n->dims[l + 1][0] = h; n->dims[l + 1][1] = w; n->dims[l + 1][2] = c;
  }
  return n;
}

void qo_net_destroy(void* nv) {
  QoNet* n = (QoNet*)nv;
  for (int l = 0; l < n->L; ++l) {
    free(n->pa[l].bias); free(n->pa[l].ctrd); free(n->pa[l].asmt); free(n->pa[l].dense);
  }
  for (int l = 0; l <= n->L; ++l) free(n->fm[l]);
  free(n->fm); free(n->dims); free(n->pa); free(n->ly); free(n);
}

int qo_net_set_params(void* nv, int l, const float* bias, const float* ctrdFile, int M, int K, int Cs,
                      const uint8_t* asmtFile) {
  QoNet* n = (QoNet*)nv;
  const QoLayer* y = &n->ly[l];
  QoParam* p = &n->pa[l];
  const int Ct = n->dims[l + 1][2];
  if (y->type != QO_CONV && y->type != QO_FCNT) return 1;
  free(p->bias); free(p->ctrd); free(p->asmt);
  p->M = M; p->K = K; p->Cs = Cs;
  p->bias = (float*)malloc(sizeof(float) * Ct);
  memcpy(p->bias, bias, sizeof(float) * Ct);
  p->ctrd = (float*)malloc(sizeof(float) * (size_t)M * K * Cs);
  qo_prep_ctrd(ctrdFile, M, K, Cs, p->ctrd);
  if (y->type == QO_CONV) {
    const size_t cnt = (size_t)Ct * y->knlSiz * y->knlSiz * M;
    p->asmt = (uint8_t*)malloc(cnt);
    qo_prep_asmt_conv(asmtFile, Ct, y->knlSiz, y->knlSiz, M, p->asmt);
  } else {
    p->asmt = (uint8_t*)malloc((size_t)Ct * M);
    qo_prep_asmt_fc(asmtFile, Ct, M, p->asmt);
  }
  return 0;
}

/* precise path (CaffePara::LoadLayerPara(false, ..), src/CaffePara.cc:290-302): bias [Ct] + conv kernels
 * [Ct][Cin/grp][kh][kw] or FC weights [Ct][D]; the layer then runs qo_conv_prec / qo_fc_prec */
int qo_net_set_dense(void* nv, int l, const float* bias, const float* weights) {
  QoNet* n = (QoNet*)nv;
  const QoLayer* y = &n->ly[l];
  QoParam* p = &n->pa[l];
  const int Ct = n->dims[l + 1][2];
  if (y->type != QO_CONV && y->type != QO_FCNT) return 1;
  const size_t cnt = (y->type == QO_CONV) ? (size_t)Ct * (n->dims[l][2] / y->grpCnt) * y->knlSiz * y->knlSiz
                                          : (size_t)Ct * n->dims[l][0] * n->dims[l][1] * n->dims[l][2];
  free(p->bias); free(p->dense);
  p->bias = (float*)malloc(sizeof(float) * Ct);
  memcpy(p->bias, bias, sizeof(float) * Ct);
  p->dense = (float*)malloc(sizeof(float) * cnt);
  memcpy(p->dense, weights, sizeof(float) * cnt);
  return 0;
}

int qo_net_fm_dims(void* nv, int l, int* hwc3) {
  QoNet* n = (QoNet*)nv;
  hwc3[0] = n->dims[l][0]; hwc3[1] = n->dims[l][1]; hwc3[2] = n->dims[l][2];
  return 0;
}

static size_t fm_elems(const QoNet* n, int l) {
  return (size_t)n->dims[l][0] * n->dims[l][1] * n->dims[l][2];
}

static void ensure_batch(QoNet* n, int B) {
  if (n->curB == B) return;
  for (int l = 0; l <= n->L; ++l) {
    free(n->fm[l]);
    n->fm[l] = (float*)malloc(sizeof(float) * fm_elems(n, l) * B);
  }
  n->curB = B;
}

/* one layer; `in` NHWC, except FC layers whose `in` is the flat vector in consumption order */
static int run_layer(QoNet* n, int l, const float* in, int B, float* out) {
  const QoLayer* y = &n->ly[l];
  const QoParam* p = &n->pa[l];
  const int H = n->dims[l][0], W = n->dims[l][1], C = n->dims[l][2];
  const int Ct = n->dims[l + 1][2];
  switch (y->type) {
    case QO_CONV: {
      if (p->dense) {
        qo_conv_prec(in, B, H, W, C, y->knlSiz, y->stride, y->padSiz, y->grpCnt, Ct, p->bias, p->dense, out);
        return 0;
      }
      if (!p->ctrd) return 2;
      float* lut = (float*)malloc(sizeof(float) * (size_t)B * H * W * p->M * p->K);
      qo_conv_aprx(in, B, H, W, C, y->knlSiz, y->stride, y->padSiz, y->grpCnt, Ct, p->bias, p->ctrd,
                   p->M, p->Cs, p->K, p->asmt, out, lut);
      free(lut);
      return 0;
    }
    case QO_FCNT: {
      if (p->dense) { qo_fc_prec(in, B, H * W * C, Ct, p->bias, p->dense, out); return 0; }
      if (!p->ctrd) return 2;
      float* lut = (float*)malloc(sizeof(float) * (size_t)B * p->M * p->K);
      qo_fc_aprx(in, B, H * W * C, Ct, p->bias, p->ctrd, p->M, p->Cs, p->K, p->asmt, out, lut);
      free(lut);
      return 0;
    }
    case QO_POOL: qo_pool(in, B, H, W, C, y->knlSiz, y->stride, y->padSiz, out); return 0;
    case QO_RELU: qo_relu(in, (int)(fm_elems(n, l) * B), out); return 0;
    case QO_LORN: qo_lrn(in, B, H, W, C, y->lrnSiz, y->lrnAlp, y->lrnBet, y->lrnIni, out); return 0;
    case QO_DRPT: memcpy(out, in, sizeof(float) * fm_elems(n, l) * B); return 0;   /* :1091-1096 */
    case QO_SMAX: qo_softmax(in, B, H * W * C, out); return 0;
    default: return 3;
  }
}

int qo_net_run_layer(void* nv, int l, const float* in, int B, float* out) {
  return run_layer((QoNet*)nv, l, in, B, out);
}

/* forward loop of src/CaffeEva.cc:213-261 (input NCHW -> NHWC :225-228 / :1146-1160; the first FC
 * layer consumes its input NCHW-flattened :187-189,236-238) */
int qo_net_forward(void* nv, const float* inNchw, int B) {
  QoNet* n = (QoNet*)nv;
  ensure_batch(n, B);
  const int H = n->inH, W = n->inW, C = n->inC;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
          n->fm[0][(((size_t)b * H + h) * W + w) * C + c] = inNchw[(((size_t)b * C + c) * H + h) * W + w];
  for (int l = 0; l < n->L; ++l) {
    const float* in = n->fm[l];
    float* tmp = NULL;
    if (l == n->firstFc && n->dims[l][0] * n->dims[l][1] > 1) {
      const int h = n->dims[l][0], w = n->dims[l][1], c = n->dims[l][2];
      tmp = (float*)malloc(sizeof(float) * fm_elems(n, l) * B);
      for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < c; ++ci)
          for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
              tmp[(((size_t)b * c + ci) * h + y) * w + x] = in[(((size_t)b * h + y) * w + x) * c + ci];
      in = tmp;
    }
    const int rc = run_layer(n, l, in, B, n->fm[l + 1]);
    free(tmp);
    if (rc) return rc;
  }
  return 0;
}

const float* qo_net_fm(void* nv, int l) { return ((QoNet*)nv)->fm[l]; }
