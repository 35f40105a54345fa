// qcnn_kernels.h — launch wrappers of the gfx950 kernels (internal; the public surface is include/qcnn_hip.h).
//
// Activation layout in HBM ("image-minor panels"): a batch is cut into panels of QCNN_PANEL = 128
// images; feature map l of one panel is a row-major matrix [E_l][128] with E_l = H*W*C elements in the
// reference's NHWC order and the 128 images of the panel innermost.  One wavefront lane carries an
// image PAIR (images 2*lane, 2*lane+1 of the panel): every load/store of a wave is a full 512-byte row,
// a table look-up is one conflict-free ds_read_b64 and the accumulation one v_pk_add_f32.  The
// code-word index of the approximate layers is wave-uniform (it depends on the layer's assignment table
// only): it is fetched as packed uint8 and broadcast into SGPRs.
#ifndef QCNN_KERNELS_H_
#define QCNN_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define QCNN_PANEL 128
#define QCNN_MAX_CS 8          // dims per sub-space supported by the LUT builders
#define QCNN_MAX_K 128         // code words per sub-space supported (a LUT stage holds 128 rows)
#define QCNN_ROWS_PAD 256      // bytes of slack after every row-index table (over-read of the last groups)
#define QCNN_STAGE_ROWS 128    // code-word rows of one LUT stage in LDS
#define QCNN_ROW_BYTES 528     // LDS row stride: 128 images * 4 B + 16 B pad (conflict-free MFMA tile writes)

// A LUT stage holds G = qcnn_stage_group(K) consecutive sub-spaces of one source pixel (conv) / of the
// input vector (FC): G * K <= 128 rows.  Assignment tables are stored on the device as uint8 ROW INDICES
// of the code word inside a stage:  row = (m % G) * K + assignment  (< 128).
static inline int qcnn_stage_group(int K) { return K <= 64 ? QCNN_STAGE_ROWS / K : 1; }
struct ConvParams {
  const float* src;      // [panels][H*W*Cin][128]
  float* dst;            // [panels][Ho*Wo*Ct][128]
  const float* bias;     // [Ct]
  const float* ctrd;     // [M][Cs][K]      (PrepCtrdBuf layout, src/CaffeEva.cc:556-557)
  const uint8_t* rows;   // [kh][kw][M][Ct] (PrepAsmtBuf layout, src/CaffeEva.cc:585-586), stage row indices
  int H, W, Cin, Ho, Wo, Ct;
  int knl, stride, pad, grp;
  int M, Cs, K;
  int relu;              // fuse max(0, x) into the store
  int panels;
  int lutF16;            // tolerance study (BASELINE configs[4]): table entries rounded to fp16 before they are stored
};

struct FcParams {
  float* partial;        // [msplit][panels][Ct][128] scratch for split-M partial sums (msplit > 1)
  int msplit;            // blocks along the sub-space axis (1 = single pass, bit-exact summation order)
  const float* src;      // [panels][D][128]
  float* dst;            // [panels][Ct][128]
  const float* bias;
  const float* ctrd;     // [M][Cs][K]
  const uint8_t* rows;   // [M][Ct]         (src/CaffeEva.cc:610-611), stage row indices
  int D, Ct, M, Cs, K;
  int relu;
  int panels;
  int lutF16;            // as ConvParams::lutF16
};

// lutMode: 0 exact VALU, 1 MFMA (2 = MFMA with fp16-rounded table entries, see lutF16).  Return hipError_t of the launch.
hipError_t qk_conv_aprx(const ConvParams& p, int lutMode, hipStream_t st);
hipError_t qk_fc_aprx(const FcParams& p, int lutMode, hipStream_t st);
int qk_fc_channels_per_block(int Ct);   // output channels one k_fc_aprx workgroup covers

// dst row e = src row map[e], rows of 128 images ([panels][D][128]); the first FC layer consumes its input
// NCHW-flattened (src/CaffeEva.cc:187-189)
hipError_t qk_permute_rows(const float* src, float* dst, const int* map, int D, int panels, hipStream_t st);
// dst[e] = sum_z partial[z][e] (z ascending), optional ReLU; n floats per partial slab
hipError_t qk_sum_partials(const float* partial, float* dst, int msplit, size_t n, int relu, hipStream_t st);

hipError_t qk_relu(const float* src, float* dst, size_t n, hipStream_t st);
hipError_t qk_lrn(const float* src, float* dst, int panels, int HW, int C, int lrnSiz, float alp, float bet,
                  float ini, hipStream_t st);
hipError_t qk_pool(const float* src, float* dst, int panels, int H, int W, int C, int Ho, int Wo, int knl,
                   int stride, int pad, hipStream_t st);
hipError_t qk_softmax(const float* src, float* dst, int panels, int C, hipStream_t st);
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
