/* TEST INFRASTRUCTURE — CPU restatement of the reference's approximate forward pass.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * it is the checker, never the product (the product path is quantized-cnn_amd/csrc + include/qcnn_hip.h
 * and fails loudly without a GPU).  Parity status: PINNED — tests/test_oracle_vs_reference.py checks
 * every function below bit-for-bit against the compiled reference (oracle/_ref/libqcnn_ref.so) in
 * the build container, and tests/golden/ holds vectors produced by that reference
 * (oracle/make_golden.py) so the pin also holds where /root/reference is absent.
 *
 * All citations are file:line into the reference tree (CAS-CLab/quantized-cnn).
 */
#ifndef QCNN_ORACLE_H_
#define QCNN_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layer type codes = ENUM_LyrType order, include/CaffePara.h:26 */
enum { QO_CONV = 0, QO_POOL = 1, QO_FCNT = 2, QO_RELU = 3, QO_LORN = 4, QO_DRPT = 5, QO_SMAX = 6 };

typedef struct {
  int type;
  int padSiz, knlSiz, knlCnt, grpCnt, stride, nodCnt, lrnSiz;
  float lrnAlp, lrnBet, lrnIni, drpRat;
} QoLayer; /* field meaning = LayerInfo, include/CaffePara.h:28-44 */

/* tolerance study only (BASELINE.json configs[4]): fp16-rounded table entries / fp16 running sums; (0, 0) = reference */
void qo_study_mode(int lutF16, int accF16);

/* ---- single functions (each mirrors one reference routine) ---- */

/* CaffeEva::GetInPdMat, src/CaffeEva.cc:1261-1296.  data [P][D], ctrd [M][Cs][K] -> lut [P][M][K] */
void qo_lut_build(const float* data, int P, int D, const float* ctrd, int M, int Cs, int K, float* lut);

/* CaffeEva::CalcFeatMap_ConvAprx, src/CaffeEva.cc:760-868.  src [B][H][W][Cin] NHWC,
 * ctrd [M][Cs][K] (already permuted), asmt [kh][kw][M][Ct] (already permuted, 0-based),
 * dst [B][Ho][Wo][Ct].  lutScratch: B*H*W*M*K floats (last group's table is left there). */
void qo_conv_aprx(const float* src, int B, int H, int W, int Cin, int knl, int stride, int pad, int grp,
                  int Ct, const float* bias, const float* ctrd, int M, int Cs, int K,
                  const uint8_t* asmt, float* dst, float* lutScratch);

/* CaffeEva::CalcFeatMap_FCntAprx, src/CaffeEva.cc:968-1025.  src [B][D], asmt [M][Ct], dst [B][Ct] */
void qo_fc_aprx(const float* src, int B, int D, int Ct, const float* bias, const float* ctrd,
                int M, int Cs, int K, const uint8_t* asmt, float* dst, float* lutScratch);

/* precise path (the reference's exact baseline, Init(false)): CalcFeatMap_ConvPrec src/CaffeEva.cc:681-758 = im2col
 * (:1195-1243) + cblas_sgemm_nn (src/BlasWrapper.cc:55-74) + bias; kn [Ct][Cin/grp][kh][kw] */
void qo_conv_prec(const float* src, int B, int H, int W, int Cin, int knl, int stride, int pad, int grp, int Ct,
                  const float* bias, const float* kn, float* dst);
/* CalcFeatMap_FCntPrec src/CaffeEva.cc:932-966 = cblas_sgemm_nt (src/BlasWrapper.cc:77-97) + bias; wei [Ct][D] */
void qo_fc_prec(const float* src, int B, int D, int Ct, const float* bias, const float* wei, float* dst);

void qo_relu(const float* src, int n, float* dst);                       /* :1027-1036 */
void qo_lrn(const float* src, int B, int H, int W, int C, int lrnSiz, float alp, float bet, float ini,
            float* dst);                                                 /* :1038-1089 + BlasWrapper.h:101-162 */
void qo_pool(const float* src, int B, int H, int W, int C, int knl, int stride, int pad, float* dst); /* :870-921 */
void qo_softmax(const float* src, int B, int C, float* dst);             /* :1098-1116 */
void qo_top5(const float* prob, int C, uint16_t* out5);                  /* :1162-1190 (destroys nothing) */
int qo_pool_out(int in, int knl, int stride, int pad);                   /* :367-370 */
int qo_conv_out(int in, int knl, int stride, int pad);                   /* :361-362 */

/* one-time layout prep */
void qo_prep_ctrd(const float* fileMKCs, int M, int K, int Cs, float* outMCsK);            /* :534-560 */
void qo_prep_asmt_conv(const uint8_t* fileCtKhKwM, int Ct, int kh, int kw, int M, uint8_t* out); /* :585-586 */
void qo_prep_asmt_fc(const uint8_t* fileCtM, int Ct, int M, uint8_t* outMCt);              /* :610-611 */

/* .cbn payload decode (include/FileIO.h:128-166); blocks = raw bytes after the header */
void qo_cbn_decode(const uint8_t* blocks, int n, int bits, uint8_t* out0based);

/* ---- whole-network runner (forward loop of src/CaffeEva.cc:213-261 for a batch) ---- */
void* qo_net_create(int inC, int inH, int inW, int layerCnt, const QoLayer* layers);
void qo_net_destroy(void* net);
/* parameters in FILE layout: bias [Ct], ctrd [M][K][Cs], asmt [Ct][kh][kw][M] or [Ct][M], 0-based */
int qo_net_set_params(void* net, int layer, const float* bias, const float* ctrdFile, int M, int K, int Cs,
                      const uint8_t* asmtFile);
/* precise path: bias [Ct] + conv kernels [Ct][Cin/grp][kh][kw] / FC weights [Ct][D] (convKnl / fcntWei file layout) */
int qo_net_set_dense(void* net, int layer, const float* bias, const float* weights);
int qo_net_fm_dims(void* net, int l, int* hwc3);
/* in: [B][C][H][W]; keeps fm[0..L] (NHWC, first-FC input kept NHWC) until the next call */
int qo_net_forward(void* net, const float* inNchw, int B);
const float* qo_net_fm(void* net, int l);
/* run one layer alone; `in` NHWC (or flat in consumption order for FC layers) */
int qo_net_run_layer(void* net, int l, const float* in, int B, float* out);

#ifdef __cplusplus
}
#endif
#endif  /* QCNN_ORACLE_H_ */
