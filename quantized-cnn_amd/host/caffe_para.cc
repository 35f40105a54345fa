// caffe_para.cc — host mirror of the reference's CaffePara (src/CaffePara.cc).
// Topology tables: src/CaffePara.cc:20-237.  Parameter loading: :239-306.  Encoding conversion: :308-358.
#include "../../include/CaffePara.h"

#include "../../include/FileIO.h"

void CaffePara::Init(const std::string& dirPathSrc, const std::string& filePfxSrc) {
  dirPath = dirPathSrc;
  filePfx = filePfxSrc;
}

// ---- table builders -----------------------------------------------------------------------------
void CaffePara::beginNet(int layers, int chn, int hei, int wid) {
  layerCnt = layers;
  imgChnIn = chn;
  imgHeiIn = hei;
  imgWidIn = wid;
  LayerInfo blank;
  memset(&blank, 0, sizeof(blank));
  blank.type = ENUM_LyrType::ReLU;
  layerInfoLst.assign(layers, blank);
  cursor_ = 0;
}

void CaffePara::addConv(int padSiz, int knlSiz, int knlCnt, int grpCnt, int stride) {
  LayerInfo& li = layerInfoLst[cursor_++];
  li.type = ENUM_LyrType::Conv;
  li.padSiz = padSiz; li.knlSiz = knlSiz; li.knlCnt = knlCnt; li.grpCnt = grpCnt; li.stride = stride;
}

void CaffePara::addPool(int padSiz, int knlSiz, int stride) {
  LayerInfo& li = layerInfoLst[cursor_++];
  li.type = ENUM_LyrType::Pool;
  li.padSiz = padSiz; li.knlSiz = knlSiz; li.stride = stride;
}

void CaffePara::addFCnt(int nodCnt) {
  LayerInfo& li = layerInfoLst[cursor_++];
  li.type = ENUM_LyrType::FCnt;
  li.nodCnt = nodCnt;
}

void CaffePara::addReLu(void) { layerInfoLst[cursor_++].type = ENUM_LyrType::ReLU; }

void CaffePara::addLoRN(int lrnSiz, float lrnAlp, float lrnBet, float lrnIni) {
  LayerInfo& li = layerInfoLst[cursor_++];
  li.type = ENUM_LyrType::LoRN;
  li.lrnSiz = lrnSiz; li.lrnAlp = lrnAlp; li.lrnBet = lrnBet; li.lrnIni = lrnIni;
}

void CaffePara::addDrpt(float drpRat) {
  LayerInfo& li = layerInfoLst[cursor_++];
  li.type = ENUM_LyrType::Drpt;
  li.drpRat = drpRat;
}

void CaffePara::addSMax(void) { layerInfoLst[cursor_++].type = ENUM_LyrType::SMax; }

// AlexNet / CaffeNet / the two fine-grained CaffeNets share one 23-layer skeleton; they differ in the
// LRN/pool order of the first two stages, the dropout ratio and the classifier width.
void CaffePara::caffeNetFamily(bool lrnBeforePool, float drpRat, int classes) {
  beginNet(23, 3, 227, 227);
  for (int stage = 0; stage < 2; ++stage) {
    if (stage == 0) addConv(0, 11, 96, 1, 4); else addConv(2, 5, 256, 2, 1);
    addReLu();
    if (lrnBeforePool) { addLoRN(5, 0.0001f, 0.75f, 1.0f); addPool(0, 3, 2); }
    else               { addPool(0, 3, 2); addLoRN(5, 0.0001f, 0.75f, 1.0f); }
  }
  addConv(1, 3, 384, 1, 1); addReLu();
  addConv(1, 3, 384, 2, 1); addReLu();
  addConv(1, 3, 256, 2, 1); addReLu();
  addPool(0, 3, 2);
  addFCnt(4096); addReLu(); addDrpt(drpRat);
  addFCnt(4096); addReLu(); addDrpt(drpRat);
  addFCnt(classes); addSMax();
}

void CaffePara::ConfigLayer_AlexNet(void) { caffeNetFamily(true, 0.50f, 1000); }       // src/CaffePara.cc:20-52
void CaffePara::ConfigLayer_CaffeNet(void) { caffeNetFamily(false, 0.50f, 1000); }     // :54-86
void CaffePara::ConfigLayer_CaffeNetFGB(void) { caffeNetFamily(false, 0.70f, 518); }   // :171-203
void CaffePara::ConfigLayer_CaffeNetFGD(void) { caffeNetFamily(false, 0.50f, 200); }   // :205-237

void CaffePara::ConfigLayer_VggCnnS(void) {                                            // :88-119
  beginNet(22, 3, 224, 224);
  addConv(0, 7, 96, 1, 2); addReLu(); addLoRN(5, 0.0005f, 0.75f, 2.0f); addPool(0, 3, 3);
  addConv(1, 5, 256, 1, 1); addReLu(); addPool(0, 2, 2);
  for (int i = 0; i < 3; ++i) { addConv(1, 3, 512, 1, 1); addReLu(); }
  addPool(0, 3, 3);
  addFCnt(4096); addReLu(); addDrpt(0.50f);
  addFCnt(4096); addReLu(); addDrpt(0.50f);
  addFCnt(1000); addSMax();
}

void CaffePara::ConfigLayer_VGG16(void) {                                              // :121-169
  beginNet(39, 3, 224, 224);
  const int reps[5] = {2, 2, 3, 3, 3};
  const int chn[5] = {64, 128, 256, 512, 512};
  for (int b = 0; b < 5; ++b) {
    for (int r = 0; r < reps[b]; ++r) { addConv(1, 3, chn[b], 1, 1); addReLu(); }
    addPool(0, 2, 2);
  }
  addFCnt(4096); addReLu(); addDrpt(0.50f);
  addFCnt(4096); addReLu(); addDrpt(0.50f);
  addFCnt(1000); addSMax();
}

// ---- parameter files ------------------------------------------------------------------------------
std::string CaffePara::layerFile(const char* kind, int layerInd, const char* ext) const {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s/%s.%s.%02d.%s", dirPath.c_str(), filePfx.c_str(), kind, layerInd + 1, ext);
  return buf;
}

bool CaffePara::LoadLayerPara(const bool enblAprx, const ENUM_AsmtEnc asmtEnc) {
  bool ok = true;
  layerParaLst.resize(layerCnt);
  for (int l = 0; l < layerCnt; ++l) {
    const LayerInfo& li = layerInfoLst[l];
    if (li.type != ENUM_LyrType::Conv && li.type != ENUM_LyrType::FCnt) continue;
    LayerPara& lp = layerParaLst[l];
    ok = FileIO::ReadBinFile(layerFile("biasVec", l, "bin"), &lp.biasVec) && ok;
    if (enblAprx) {
      ok = FileIO::ReadBinFile(layerFile("ctrdLst", l, "bin"), &lp.ctrdLst) && ok;
      bool got;
      if (asmtEnc == ENUM_AsmtEnc::Raw) got = FileIO::ReadBinFile(layerFile("asmtLst", l, "bin"), &lp.asmtLst);
      else got = FileIO::ReadCbnFile(layerFile("asmtLst", l, "cbn"), &lp.asmtLst);
      ok = got && ok;
      if (got) {   // files hold MATLAB-style 1-based indices (src/CaffePara.cc:285-288)
        uint8_t* a = lp.asmtLst.GetDataPtr();
        const int n = lp.asmtLst.GetEleCnt();
        for (int i = 0; i < n; ++i) a[i] = static_cast<uint8_t>(a[i] - 1);
      }
    } else if (li.type == ENUM_LyrType::Conv) {
      ok = FileIO::ReadBinFile(layerFile("convKnl", l, "bin"), &lp.convKnlLst) && ok;
    } else {
      ok = FileIO::ReadBinFile(layerFile("fcntWei", l, "bin"), &lp.fcntWeiMat) && ok;
    }
  }
  return ok;
}

bool CaffePara::CvtAsmtEnc(const ENUM_AsmtEnc asmtEncSrc, const ENUM_AsmtEnc asmtEncDst) {
  if (asmtEncSrc == asmtEncDst) {
    printf("[INFO] no encoding conversion is required\n");
    return true;
  }
  bool ok = true;
  Matrix<uint8_t> idx;   // 1-based, as stored by the Raw files and returned by ReadCbnFile
  for (int l = 0; l < layerCnt; ++l) {
    const LayerInfo& li = layerInfoLst[l];
    if (li.type != ENUM_LyrType::Conv && li.type != ENUM_LyrType::FCnt) continue;
    const std::string rawPath = layerFile("asmtLst", l, "bin");
    const std::string cbnPath = layerFile("asmtLst", l, "cbn");
    if (asmtEncSrc == ENUM_AsmtEnc::Raw) {
      if (!FileIO::ReadBinFile(rawPath, &idx)) { ok = false; continue; }
      const int bits = CalcBitCntPerEle(idx);
      printf("layer #%d: bitCntPerEle = %d\n", l + 1, bits);
      ok = FileIO::WriteCbnFile(cbnPath, idx, bits) && ok;
    } else {
      if (!FileIO::ReadCbnFile(cbnPath, &idx)) { ok = false; continue; }
      ok = FileIO::WriteBinFile(rawPath, idx) && ok;
    }
  }
  return ok;
}

// bits needed for the largest stored value (index - 1), src/CaffePara.cc:360-380
int CaffePara::CalcBitCntPerEle(const Matrix<uint8_t>& asmtLst) {
  const uint8_t* a = asmtLst.GetDataPtr();
  const int n = asmtLst.GetEleCnt();
  unsigned top = 0;
  for (int i = 0; i < n; ++i) top = std::max<unsigned>(top, a[i]);
  if (top > 0) top -= 1;
  int bits = 0;
  for (; top != 0; top >>= 1) ++bits;
  return bits;
}
