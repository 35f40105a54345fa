// CaffePara.h — network topology tables and per-layer parameter loading, with the public surface of
// the reference's CaffePara (include/CaffePara.h:54-91): same method names, same PUBLIC DATA MEMBERS
// (dirPath, filePfx, layerCnt, imgChnIn, imgHeiIn, imgWidIn, layerInfoLst, layerParaLst), same
// LayerInfo / LayerPara field names, same enum values — callers written against the reference compile
// unchanged.  Host-side only; CaffeEva hands the loaded tensors to the device through include/qcnn_hip.h.
#ifndef QCNN_HOST_CAFFEPARA_H_
#define QCNN_HOST_CAFFEPARA_H_

#include <string>
#include <vector>

#include "../include/Common.h"
#include "../include/Matrix.h"

// How the assignment (code-word index) files are stored on disk:
//   Raw      <pfx>.asmtLst.NN.bin  one byte per index
//   Compact  <pfx>.asmtLst.NN.cbn  bit-packed, see FileIO.h
enum class ENUM_AsmtEnc {Raw, Compact};

// The numeric order is part of the C-ABI (QCNN_CONV .. QCNN_SMAX in qcnn_hip.h).
enum class ENUM_LyrType {Conv, Pool, FCnt, ReLU, LoRN, Drpt, SMax};

typedef struct {
  ENUM_LyrType type;
  int padSiz;     // zero padding on each border (conv / pool)
  int knlSiz;     // square window edge (conv / pool)
  int knlCnt;     // output channels of a conv layer
  int grpCnt;     // channel groups of a conv layer
  int stride;     // spatial step (conv / pool)
  int nodCnt;     // output neurons of a fully-connected layer
  int lrnSiz;     // LRN window across channels
  float lrnAlp;   // LRN alpha
  float lrnBet;   // LRN beta
  float lrnIni;   // LRN additive constant k
  float drpRat;   // dropout keep ratio (a no-op at test time)
} LayerInfo;
typedef std::vector<LayerInfo> LayerInfoLst;

typedef struct {
  Matrix<float> convKnlLst;   // exact conv kernels   (precise path only; never shipped)
  Matrix<float> fcntWeiMat;   // exact FC weights     (precise path only; never shipped)
  Matrix<float> biasVec;      // [Ct]
  Matrix<float> ctrdLst;      // sub-codebooks, file order [M][K][Cs]
  Matrix<uint8_t> asmtLst;    // code-word indices, 0-based after LoadLayerPara: [Ct][kh][kw][M] / [Ct][M]
} LayerPara;
typedef std::vector<LayerPara> LayerParaLst;

class CaffePara {
 public:
  void Init(const std::string& dirPathSrc, const std::string& filePfxSrc);
  void ConfigLayer_AlexNet(void);
  void ConfigLayer_CaffeNet(void);
  void ConfigLayer_VggCnnS(void);
  void ConfigLayer_VGG16(void);
  void ConfigLayer_CaffeNetFGB(void);
  void ConfigLayer_CaffeNetFGD(void);
  // Reads <dir>/<pfx>.{biasVec,ctrdLst}.NN.bin and .asmtLst.NN.{bin|cbn} for every conv/FC layer
  // (NN = 1-based layer index) and shifts the stored 1-based indices to 0-based.
  bool LoadLayerPara(const bool enblAprx, const ENUM_AsmtEnc asmtEnc);
  // Re-encodes every layer's assignment file Raw -> Compact or Compact -> Raw.
  bool CvtAsmtEnc(const ENUM_AsmtEnc asmtEncSrc, const ENUM_AsmtEnc asmtEncDst);

 public:
  std::string dirPath;
  std::string filePfx;
  int layerCnt;
  int imgChnIn;
  int imgHeiIn;
  int imgWidIn;
  LayerInfoLst layerInfoLst;
  LayerParaLst layerParaLst;

 private:
  int CalcBitCntPerEle(const Matrix<uint8_t>& asmtLst);
  void beginNet(int layers, int chn, int hei, int wid);
  void addConv(int padSiz, int knlSiz, int knlCnt, int grpCnt, int stride);
  void addPool(int padSiz, int knlSiz, int stride);
  void addFCnt(int nodCnt);
  void addReLu(void);
  void addLoRN(int lrnSiz, float lrnAlp, float lrnBet, float lrnIni);
  void addDrpt(float drpRat);
  void addSMax(void);
  void caffeNetFamily(bool lrnBeforePool, float drpRat, int classes);
  std::string layerFile(const char* kind, int layerInd, const char* ext) const;
  int cursor_;
};

#endif  // QCNN_HOST_CAFFEPARA_H_
