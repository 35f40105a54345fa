// TEST INFRASTRUCTURE — never linked into, or called by, the product path.
//
// Thin C-ABI around the *unmodified* reference implementation (CAS-CLab/quantized-cnn), compiled
// from the sources where they lie (REF=/root/reference) by oracle/Makefile into
// oracle/_ref/libqcnn_ref.so.  It exists so that pytest (ctypes) can
//   * run the reference's own CaffeEva::ExecForwardPass(img, prob) (src/CaffeEva.cc:213-261),
//   * read every featMapLst[l] and every look-up table the reference produced,
//   * run ONE layer of the reference in isolation on a caller-supplied input,
//   * time the reference on the host cores (bench.py cpu_baseline, kind "reference").
// Nothing of the reference is copied here: private members are reached with the
// `#define private public` trick, the arithmetic stays in the reference objects.
//
// Only oracle/Makefile builds this file; it needs -I$(REF)/include.

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <float.h>
#include <time.h>
#include <string>
#include <vector>
#include <algorithm>
#include <iostream>
#include <typeinfo>
#include <chrono>

#define private public
#include "CaffeEvaWrapper.h"   // resolved through -I$(REF)/include; pulls CaffeEva.h, CaffePara.h, Matrix.h
#undef private

namespace {

struct RefHandle {
  CaffeEva* eva;
  bool loaded;
  int firstFc;
};

int find_first_fc(const CaffePara& p) {
  for (int l = 0; l < p.layerCnt; ++l)
    if (p.layerInfoLst[l].type == ENUM_LyrType::FCnt) return l;
  return -1;
}

// After ExecForwardPass the input of the first FC layer holds NCHW-ordered data relabelled as
// [N,H,W,C] (src/CaffeEva.cc:187-204).  Undo that so callers always see NHWC.
void copy_fm_nhwc(const RefHandle* h, int l, float* out) {
  const Matrix<float>& fm = h->eva->featMapLst[l];
  const int n = fm.GetDimLen(0), hh = fm.GetDimLen(1), ww = fm.GetDimLen(2), cc = fm.GetDimLen(3);
  const float* src = fm.GetDataPtr();
  if (l != h->firstFc || hh * ww == 1) {
    memcpy(out, src, sizeof(float) * fm.GetEleCnt());
    return;
  }
  for (int in = 0; in < n; ++in)
    for (int c = 0; c < cc; ++c)
      for (int y = 0; y < hh; ++y)
        for (int x = 0; x < ww; ++x)
          out[((in * hh + y) * ww + x) * cc + c] = src[((in * cc + c) * hh + y) * ww + x];
}

}  // namespace

extern "C" {

void* qref_create(void) {
  RefHandle* h = new RefHandle;
  h->eva = reinterpret_cast<CaffeEva*>(::operator new(sizeof(CaffeEva)));
  memset(static_cast<void*>(h->eva), 0, sizeof(CaffeEva));   // the class has no ctor
  new (h->eva) CaffeEva;
  h->loaded = false;
  h->firstFc = -1;
  return h;
}

// The reference destructor walks state that only exists after a successful load; leak otherwise.
void qref_destroy(void* hv) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  if (h->loaded) delete h->eva;
  delete h;
}

int qref_load_named(void* hv, const char* model, const char* dir, const char* pfx) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  h->eva->Init(true);
  h->eva->SetModelName(model);
  h->eva->SetModelPath(dir, pfx);
  if (!h->eva->LoadCaffePara()) return 1;
  h->loaded = true;
  h->firstFc = find_first_fc(h->eva->caffeParaObj);
  return 0;
}

// Arbitrary topology through the reference's own loaders / buffer planners.
// iparams[l] = {pad, knl, cnt, grp, stride, nod, lrnSiz}; fparams[l] = {lrnAlp, lrnBet, lrnIni, drpRat}
static int load_custom(void* hv, const char* dir, const char* pfx, int inC, int inH, int inW,
                       int layerCnt, const int* types, const int* iparams, const float* fparams, bool aprx);

int qref_load_custom(void* hv, const char* dir, const char* pfx, int inC, int inH, int inW,
                     int layerCnt, const int* types, const int* iparams, const float* fparams) {
  return load_custom(hv, dir, pfx, inC, inH, inW, layerCnt, types, iparams, fparams, true);
}

// The same with the reference's PRECISE path (Init(false): convKnl / fcntWei files, im2col + sgemm).
int qref_load_custom_prec(void* hv, const char* dir, const char* pfx, int inC, int inH, int inW,
                          int layerCnt, const int* types, const int* iparams, const float* fparams) {
  return load_custom(hv, dir, pfx, inC, inH, inW, layerCnt, types, iparams, fparams, false);
}

static int load_custom(void* hv, const char* dir, const char* pfx, int inC, int inH, int inW,
                       int layerCnt, const int* types, const int* iparams, const float* fparams, bool aprx) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  CaffeEva& e = *h->eva;
  e.Init(aprx);
  e.SetModelName("custom");
  e.SetModelPath(dir, pfx);
  CaffePara& p = e.caffeParaObj;
  p.Init(dir, pfx);
  p.layerCnt = layerCnt;
  p.imgChnIn = inC;
  p.imgHeiIn = inH;
  p.imgWidIn = inW;
  p.layerInfoLst.resize(layerCnt);
  for (int l = 0; l < layerCnt; ++l) {
    LayerInfo& li = p.layerInfoLst[l];
    memset(&li, 0, sizeof(li));
    li.type = static_cast<ENUM_LyrType>(types[l]);
    const int* ip = iparams + 7 * l;
    const float* fp = fparams + 4 * l;
    li.padSiz = ip[0]; li.knlSiz = ip[1]; li.knlCnt = ip[2]; li.grpCnt = ip[3];
    li.stride = ip[4]; li.nodCnt = ip[5]; li.lrnSiz = ip[6];
    li.lrnAlp = fp[0]; li.lrnBet = fp[1]; li.lrnIni = fp[2]; li.drpRat = fp[3];
  }
  if (!p.LoadLayerPara(aprx, ENUM_AsmtEnc::Compact)) return 1;
  e.PrepFeatMap();
  e.PrepFeatBuf();
  if (aprx) {                      // as CaffeEva::LoadCaffePara, src/CaffeEva.cc:141-146
    e.PrepCtrdBuf();
    e.PrepAsmtBuf();
  }
  h->loaded = true;
  h->firstFc = find_first_fc(p);
  return 0;
}

int qref_layer_cnt(void* hv) {
  return static_cast<RefHandle*>(hv)->eva->caffeParaObj.layerCnt;
}

int qref_fm_dims(void* hv, int l, int* dims4) {
  const FeatMapSiz& s = static_cast<RefHandle*>(hv)->eva->featMapSizLst[l];
  dims4[0] = s.dataCnt; dims4[1] = s.imgHei; dims4[2] = s.imgWid; dims4[3] = s.imgChn;
  return 0;
}

// One image, NCHW fp32 in, probabilities out — the reference's own single-image entry point.
int qref_forward(void* hv, const float* imgNchw, float* prob) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  const CaffePara& p = h->eva->caffeParaObj;
  Matrix<float> img(1, p.imgChnIn, p.imgHeiIn, p.imgWidIn);
  memcpy(img.GetDataPtr(), imgNchw, sizeof(float) * img.GetEleCnt());
  Matrix<float> probVec;
  h->eva->ExecForwardPass(img, &probVec);
  if (prob) memcpy(prob, probVec.GetDataPtr(), sizeof(float) * probVec.GetEleCnt());
  return 0;
}

int qref_get_fm(void* hv, int l, float* out) {
  copy_fm_nhwc(static_cast<RefHandle*>(hv), l, out);
  return 0;
}

// Look-up table of layer l as left behind by the last forward pass ([P, M, K]; for grouped conv
// layers it is the LAST group's table, the buffer is reused per group, src/CaffeEva.cc:784,810).
int qref_lut_elems(void* hv, int l) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  const ENUM_LyrType t = h->eva->caffeParaObj.layerInfoLst[l].type;
  if (t == ENUM_LyrType::Conv) return h->eva->featBufStrMat[l][4].pFeatBuf->GetEleCnt();
  if (t == ENUM_LyrType::FCnt) return h->eva->featBufStrMat[l][1].pFeatBuf->GetEleCnt();
  return 0;
}

int qref_get_lut(void* hv, int l, float* out) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  const ENUM_LyrType t = h->eva->caffeParaObj.layerInfoLst[l].type;
  const Matrix<float>* m = nullptr;
  if (t == ENUM_LyrType::Conv) m = h->eva->featBufStrMat[l][4].pFeatBuf;
  if (t == ENUM_LyrType::FCnt) m = h->eva->featBufStrMat[l][1].pFeatBuf;
  if (!m) return 1;
  memcpy(out, m->GetDataPtr(), sizeof(float) * m->GetEleCnt());
  return 0;
}

// Run layer l of the reference alone: `in` is copied verbatim into featMapLst[l] (NHWC for
// spatial layers; for FC layers the flat vector in the order the reference consumes it, i.e.
// NCHW-flattened for the first FC), then the reference's private dispatcher is called.
int qref_run_layer(void* hv, int l, const float* in, float* out) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  Matrix<float>& src = h->eva->featMapLst[l];
  Matrix<float>& dst = h->eva->featMapLst[l + 1];
  memcpy(src.GetDataPtr(), in, sizeof(float) * src.GetEleCnt());
  h->eva->CalcFeatMap(src, l, &dst);
  memcpy(out, dst.GetDataPtr(), sizeof(float) * dst.GetEleCnt());
  return 0;
}

// Top-5 of the last feature map through the reference's own argmax (src/CaffeEva.cc:1162-1190).
int qref_top5(void* hv, uint16_t* out5) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  Matrix<uint16_t> labl(1, 5, 1, 1);
  h->eva->CvtFeatMapToLablVec(0, 0, h->eva->featMapLst[h->eva->caffeParaObj.layerCnt], &labl);
  for (int i = 0; i < 5; ++i) out5[i] = labl.GetEleAt(0, i, 0, 0);
  return 0;
}

// Time n single-image forward passes (the reference's batch-1 regime, src/CaffeEva.cc:23).
// Returns wall seconds; *cpuSeconds receives the reference's own swAllLayers reading
// (clock()-based CPU time summed over CalcFeatMap calls, include/StopWatch.h:45,52).
double qref_time_forward(void* hv, const float* imgsNchw, int n, double* cpuSeconds) {
  RefHandle* h = static_cast<RefHandle*>(hv);
  const CaffePara& p = h->eva->caffeParaObj;
  const size_t per = static_cast<size_t>(p.imgChnIn) * p.imgHeiIn * p.imgWidIn;
  Matrix<float> img(1, p.imgChnIn, p.imgHeiIn, p.imgWidIn);
  Matrix<float> probVec;
  h->eva->Init(true);   // resets the stop-watches only
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) {
    memcpy(img.GetDataPtr(), imgsNchw + per * i, sizeof(float) * per);
    h->eva->ExecForwardPass(img, &probVec);
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (cpuSeconds) *cpuSeconds = h->eva->swAllLayers.GetTime();
  return std::chrono::duration<double>(t1 - t0).count();
}

// BMP -> network input through the reference's own BmpImgIO (src/BmpImgIO.cc:40-71) with the
// AlexNet settings of CaffeEvaWrapper::SetModel (src/CaffeEvaWrapper.cc:62-67).  out: [3][crop][crop]
int qref_load_bmp(const char* meanPath, const char* bmpPath, int full, int crop, float* out) {
  BmpImgIOPara para;
  para.reszType = ENUM_ReszType::Strict;
  para.meanType = ENUM_MeanType::Full;
  para.imgHeiFull = full; para.imgWidFull = full;
  para.imgHeiCrop = crop; para.imgWidCrop = crop;
  para.filePathMean = meanPath;
  BmpImgIO io;
  if (!io.Init(para)) return 1;
  Matrix<float> img;
  if (!io.Load(bmpPath, &img)) return 2;
  memcpy(out, img.GetDataPtr(), sizeof(float) * img.GetEleCnt());
  return 0;
}

}  // extern "C"
