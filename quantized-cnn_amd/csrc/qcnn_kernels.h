// qcnn_kernels.h — launch wrappers of the gfx950 kernels (internal; the public surface is include/qcnn_hip.h).
//
// Activation layout in HBM ("image-minor panels"): a batch is cut into panels of QCNN_PANEL = 64
// images; feature map l of one panel is a row-major matrix [E_l][64] with E_l = H*W*C elements in the
// reference's NHWC order and the 64 images of the panel innermost.  One wavefront lane = one image:
// every load/store of a wave is a full 256-byte row, and the code-word index of the approximate
// layers is wave-uniform (it depends on the layer's assignment table only), so table look-ups become
// conflict-free LDS row reads instead of random gathers.
#ifndef QCNN_KERNELS_H_
#define QCNN_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define QCNN_PANEL 64
#define QCNN_MAX_CS 8          // dims per sub-space supported by the LUT builders
#define QCNN_ASMT_PAD 64       // bytes of slack after every assignment table (vector over-read)

struct ConvParams {
  const float* src;      // [panels][H*W*Cin][64]
  float* dst;            // [panels][Ho*Wo*Ct][64]
  const float* bias;     // [Ct]
  const float* ctrd;     // [M][Cs][K]      (PrepCtrdBuf layout, src/CaffeEva.cc:556-557)
  const uint8_t* asmt;   // [kh][kw][M][Ct] (PrepAsmtBuf layout, src/CaffeEva.cc:585-586)
  int H, W, Cin, Ho, Wo, Ct;
  int knl, stride, pad, grp;
  int M, Cs, K;
  int relu;              // fuse max(0, x) into the store
  int panels;
};

struct FcParams {
  const float* src;      // [panels][D][64]
  float* dst;            // [panels][Ct][64]
  const float* bias;
  const float* ctrd;     // [M][Cs][K]
  const uint8_t* asmt;   // [M][Ct]         (src/CaffeEva.cc:610-611)
  const int* dmap;       // [D] row of input element d in src (NCHW-flatten of the first FC), or NULL
  int D, Ct, M, Cs, K;
  int relu;
  int panels;
};

// lutMode: 0 exact VALU, 1 MFMA.  Return hipError_t of the launch.
hipError_t qk_conv_aprx(const ConvParams& p, int lutMode, hipStream_t st);
hipError_t qk_fc_aprx(const FcParams& p, int lutMode, hipStream_t st);

hipError_t qk_relu(const float* src, float* dst, size_t n, hipStream_t st);
hipError_t qk_lrn(const float* src, float* dst, int panels, int HW, int C, int lrnSiz, float alp, float bet,
                  float ini, hipStream_t st);
hipError_t qk_pool(const float* src, float* dst, int panels, int H, int W, int C, int Ho, int Wo, int knl,
                   int stride, int pad, hipStream_t st);
hipError_t qk_softmax(const float* src, float* dst, int panels, int C, hipStream_t st);
hipError_t qk_top5(const float* prob, uint16_t* out, int n, int C, hipStream_t st);   // prob panel layout -> [n][5]

// [n][C][H][W] -> panels [H*W*C][64] (lanes >= n zero-filled)
hipError_t qk_pack_nchw(const float* in, float* dst, int n, int C, int H, int W, hipStream_t st);
// [n][E] (already in NHWC / flat order) -> panels [E][64]
hipError_t qk_pack_rows(const float* in, float* dst, int n, int E, hipStream_t st);
// panels [E][64] -> [n][E]
hipError_t qk_unpack_rows(const float* src, float* out, int n, int E, hipStream_t st);

#endif  // QCNN_KERNELS_H_
