// CaffeEva.h — the inference engine behind the reference's public interface (include/CaffeEva.h:64-85),
// re-implemented for MI355X: the ten public methods keep their names, arguments and return
// conventions (bool success / printf("[ERROR] ...")), so src/Main.cc + src/UnitTest.cc of the reference
// drive it unchanged.  Everything private is different: there are no host feature maps or buffers —
// the object owns one device group (include/qcnn_hip.h: a context per GPU) and all arithmetic of the
// approximate forward pass runs there.
//
// Behavioural notes
//  * Init(true) = the approximate path (the product); Init(false) = the reference's precise path (dense conv kernels /
//    FC weights from convKnl.NN.bin / fcntWei.NN.bin, which the reference never shipped) as an exact baseline on the
//    device (qcnn_model_set_layer_dense / _weights).
//  * The reference hard-codes 1 image per batch and 100 batches (src/CaffeEva.cc:23-24).  Here they are
//    the DEFAULTS of two environment variables read at LoadCaffePara():
//        QCNN_BATCH    images per forward pass       (default 1)
//        QCNN_BATCHES  forward passes to run         (default 100)
//        QCNN_DEVICES  "all" (default) | "0,1,.."    GPUs the batch is sharded over (RCCL parameter broadcast)
//        QCNN_DEVICE   one HIP device ordinal        (shorthand for a one-GPU group)
//        QCNN_MAX_INFLIGHT  images per device batch  (default 1024): ExecForwardPass(void) hands the
//                      QCNN_BATCHES x QCNN_BATCH images to the GPUs in chunks of that size — the same images,
//                      prints and results as one batch at a time (QCNN_COALESCE=0 restores that)
//        QCNN_LUT      "mfma" (default) | "exact"    look-up-table builder (exact = bit-identical conv/FC)
//        QCNN_KEEP_ALL 0 (default) | 1               1 = every layer writes its own feature map (GetFeatMap() of maps the
//                      fast path fuses away: pre-ReLU conv/FC outputs, LRN maps in front of a pool)
//        QCNN_WARMUP   1 (default) | 0               LoadCaffePara() ends with two one-panel forward passes on zeros
//                      (code objects loaded, staging buffers and streams created before the first timed pass)
//        QCNN_PIN_DATASET copy (default) | register | 0   where the image block lives: a pinned buffer of the HIP runtime
//                      (uploads are DMA transfers at the full PCIe rate, overlapped with the previous batch's layers),
//                      dataLst's own storage registered with the runtime (no second copy, about half the rate), or
//                      plain pageable memory
//  * ExecForwardPass(void) hands ALL its device batches to the group in one call: the upload of a batch runs under
//    the layers of the one before it; every image goes through the panel kernels.  ExecForwardPass(img, prob) is the
//    latency mode: the few-image kernels (results equal to rounding, ~1e-6).
//  * DispElpsTime() prints the reference's stop-watch names; the values are HIP-event times of the
//    layers (LUT build and look-up are one fused kernel, so swCompLkupTbl* report 0 and swEstiInPdVal*
//    carry the fused time).
#ifndef QCNN_HOST_CAFFEEVA_H_
#define QCNN_HOST_CAFFEEVA_H_

#include <string>
#include <vector>

#include "../include/Common.h"
#include "../include/BlasWrapper.h"
#include "../include/CaffePara.h"
#include "../include/Matrix.h"
#include "../include/StopWatch.h"

struct QcnnCtx;     // include/qcnn_hip.h
struct QcnnGroup;

class CaffeEva {
 public:
  CaffeEva(void);
  ~CaffeEva(void);

 public:
  void Init(const bool enblAprxSrc);
  void SetModelName(const std::string& modelNameSrc);
  void SetModelPath(const std::string& dirPathMainSrc, const std::string& fileNamePfxSrc);
  bool LoadDataset(const std::string& dirPathData);
  bool LoadCaffePara(void);
  // classify QCNN_BATCHES x QCNN_BATCH images of the loaded dataset
  void ExecForwardPass(void);
  // one image [1, C, H, W] in, class probabilities out
  void ExecForwardPass(const Matrix<float>& imgDataIn, Matrix<float>* pProbVecOut);
  void CalcPredAccu(void);
  float DispElpsTime(void);

  // extensions (not in the reference): feature map l of the last forward pass, NHWC per image
  bool GetFeatMap(const int layerInd, const int dataCnt, Matrix<float>* pFeatMap);
  std::string GetErrorMsg(void) const { return lastError_; }

 private:
  CaffeEva(const CaffeEva&);
  CaffeEva& operator=(const CaffeEva&);

  bool enblAprx;
  std::string modelName;
  std::string dirPathMain;
  std::string fileNamePfx;
  CaffePara caffeParaObj;
  Matrix<float> dataLst;            // [N, C, H, W]
  Matrix<uint16_t> lablVecGrth;     // ground truth, 0-based
  Matrix<uint16_t> lablVecPred;     // [N, 5]

  QcnnGroup* grp_;                  // device group (one context per GPU + RCCL communicator); owns every device buffer
  QcnnCtx* ctx_;                    // rank 0 of the group: single-image passes, feature-map dumps, timers
  bool modelReady_;
  int batchSize_;                   // QCNN_BATCH
  int batchCnt_;                    // QCNN_BATCHES
  int inflight_;                    // images handed to the device group at once
  int imagesDone_;                  // images classified by the last ExecForwardPass(void)
  bool keepAll_;                    // QCNN_KEEP_ALL: layer-for-layer mode (every feature map materialised)
  float* pinned_;                   // dataLst storage registered with the HIP runtime (qcnn_host_register), or NULL
  float* pinnedCopy_;               // the images in a pinned buffer of the HIP runtime (qcnn_host_alloc), or NULL
  int pinnedImages_;                // leading images of the dataset the pinned buffer / registration covers
  std::string lastError_;
  StopWatch swWall_;                // wall clock around the forward passes (host view)

  bool fail(const std::string& what);
  bool buildDeviceModel(void);
  void pinDataset(void);
  void unpinDataset(void);
};

#endif  // QCNN_HOST_CAFFEEVA_H_
