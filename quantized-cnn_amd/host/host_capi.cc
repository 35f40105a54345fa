// host_capi.cc — small C entry points into the C++ host mirror, for the pytest suite only (ctypes cannot
// call C++ classes).  Nothing here is on the product path.
#include "../../include/BmpImgIO.h"
#ifndef QH_NO_DEVICE          // the sanitizer build (tests/test_host_sanitizers.py) covers the device-free classes only
#include "../../include/CaffeEva.h"
#endif
#include "../../include/CaffePara.h"
#include "../../include/FileIO.h"
#include "../../include/Matrix.h"

extern "C" {

// BmpImgIO::Init + Load; out [3][crop][crop]
int qh_bmp_load(const char* meanPath, const char* bmpPath, int full, int crop, int relaxed, float* out) {
  BmpImgIOPara para;
  para.reszType = relaxed ? ENUM_ReszType::Relaxed : ENUM_ReszType::Strict;
  para.meanType = relaxed ? ENUM_MeanType::Crop : ENUM_MeanType::Full;
  para.imgHeiFull = full; para.imgWidFull = full; para.imgHeiCrop = crop; para.imgWidCrop = crop;
  para.filePathMean = meanPath;
  BmpImgIO io;
  if (!io.Init(para)) return 1;
  Matrix<float> img;
  if (!io.Load(bmpPath, &img)) return 2;
  memcpy(out, img.GetDataPtr(), sizeof(float) * img.GetEleCnt());
  return 0;
}

// CaffePara: configure a named model, load its parameters; report layer count and per-layer sizes
int qh_para_load(const char* model, const char* dir, const char* pfx, int raw, int* layerCnt, int* dims /*[L][8]*/,
                 double* sums /*[L][3]*/) {
  CaffePara p;
  p.Init(dir, pfx);
  const std::string m = model;
  if (m == "AlexNet") p.ConfigLayer_AlexNet();
  else if (m == "CaffeNet") p.ConfigLayer_CaffeNet();
  else if (m == "VggCnnS") p.ConfigLayer_VggCnnS();
  else if (m == "VGG16") p.ConfigLayer_VGG16();
  else if (m == "CaffeNetFGB") p.ConfigLayer_CaffeNetFGB();
  else if (m == "CaffeNetFGD") p.ConfigLayer_CaffeNetFGD();
  else return 1;
  *layerCnt = p.layerCnt;
  for (int l = 0; l < p.layerCnt; ++l) {
    const LayerInfo& li = p.layerInfoLst[l];
    int* d = dims + 8 * l;
    d[0] = static_cast<int>(li.type); d[1] = li.padSiz; d[2] = li.knlSiz; d[3] = li.knlCnt;
    d[4] = li.grpCnt; d[5] = li.stride; d[6] = li.nodCnt; d[7] = li.lrnSiz;
  }
  if (dir[0] == '\0') return 0;   // topology only
  if (!p.LoadLayerPara(true, raw ? ENUM_AsmtEnc::Raw : ENUM_AsmtEnc::Compact)) return 2;
  for (int l = 0; l < p.layerCnt; ++l) {
    const LayerPara& lp = p.layerParaLst[l];
    double sb = 0, sc = 0, sa = 0;
    for (int i = 0; i < lp.biasVec.GetEleCnt(); ++i) sb += lp.biasVec.GetDataPtr()[i];
    for (int i = 0; i < lp.ctrdLst.GetEleCnt(); ++i) sc += lp.ctrdLst.GetDataPtr()[i];
    for (int i = 0; i < lp.asmtLst.GetEleCnt(); ++i) sa += lp.asmtLst.GetDataPtr()[i];
    sums[3 * l] = sb; sums[3 * l + 1] = sc; sums[3 * l + 2] = sa;
  }
  return 0;
}

// CaffePara::CvtAsmtEnc on an AlexNet-shaped directory
int qh_para_convert(const char* dir, const char* pfx, int toCompact) {
  CaffePara p;
  p.Init(dir, pfx);
  p.ConfigLayer_AlexNet();
  return p.CvtAsmtEnc(toCompact ? ENUM_AsmtEnc::Raw : ENUM_AsmtEnc::Compact,
                      toCompact ? ENUM_AsmtEnc::Compact : ENUM_AsmtEnc::Raw) ? 0 : 1;
}

// FileIO .cbn: read (1-based) -> write -> compare; returns 0 when identical
int qh_cbn_rewrite(const char* inPath, const char* outPath, int bits) {
  Matrix<uint8_t> m;
  if (!FileIO::ReadCbnFile(inPath, &m)) return 1;
  if (!FileIO::WriteCbnFile(outPath, m, bits)) return 2;
  return 0;
}

#ifndef QH_NO_DEVICE
// CaffeEva through its public interface on one BMP image (AlexNet, data root laid out like the reference's):
// feature maps `layers[0..n)` of the forward pass, NHWC, written back to back into out (cap floats); sizes[] gets
// the element counts.  Arithmetic sanity of the host mirror -> C-ABI plumbing (feature maps, not accuracy lines).
int qh_eva_featmaps(const char* mainDir, const char* bmpPath, const int* layers, int n, float* out, int cap, int* sizes) {
  BmpImgIOPara para;
  para.reszType = ENUM_ReszType::Strict;
  para.meanType = ENUM_MeanType::Full;
  para.imgHeiFull = 256; para.imgWidFull = 256; para.imgHeiCrop = 227; para.imgWidCrop = 227;
  para.filePathMean = std::string(mainDir) + "/AlexNet/imagenet_mean.single.bin";
  BmpImgIO io;
  if (!io.Init(para)) return 1;
  Matrix<float> img;
  if (!io.Load(bmpPath, &img)) return 2;
  CaffeEva eva;
  eva.Init(true);
  eva.SetModelName("AlexNet");
  eva.SetModelPath(std::string(mainDir) + "/AlexNet/Bin.Files", "bvlc_alexnet_aCaF");
  if (!eva.LoadCaffePara()) return 3;
  Matrix<float> prob;
  eva.ExecForwardPass(img, &prob);
  if (!eva.GetErrorMsg().empty()) return 4;
  int used = 0;
  for (int i = 0; i < n; ++i) {
    Matrix<float> fm;
    if (!eva.GetFeatMap(layers[i], 1, &fm)) return 5;
    if (used + fm.GetEleCnt() > cap) return 6;
    memcpy(out + used, fm.GetDataPtr(), sizeof(float) * fm.GetEleCnt());
    sizes[i] = fm.GetEleCnt();
    used += fm.GetEleCnt();
  }
  return 0;
}

// CaffeEva with Init(aprx) on a data root laid out like the reference's, one image [3][227][227] in, probabilities out
int qh_eva_prob(const char* mainDir, const char* model, const char* sub, const char* pfx, int aprx, const float* img,
                int c, int h, int w, float* prob, int cap) {
  CaffeEva eva;
  eva.Init(aprx != 0);
  eva.SetModelName(model);
  eva.SetModelPath(std::string(mainDir) + "/" + sub, pfx);
  if (!eva.LoadCaffePara()) return 3;
  Matrix<float> in(1, c, h, w), out;
  memcpy(in.GetDataPtr(), img, sizeof(float) * in.GetEleCnt());
  eva.ExecForwardPass(in, &out);
  if (!eva.GetErrorMsg().empty()) return 4;
  if (out.GetEleCnt() > cap) return 6;
  memcpy(prob, out.GetDataPtr(), sizeof(float) * out.GetEleCnt());
  return out.GetEleCnt() > 0 ? 0 : 7;
}

#endif  // QH_NO_DEVICE

// Matrix semantics the reference relies on; returns the number of failed checks
int qh_matrix_selftest(void) {
  int bad = 0;
  Matrix<float> a(2, 3, 4, 5);
  for (int i = 0; i < a.GetEleCnt(); ++i) a.GetDataPtr()[i] = static_cast<float>(i);
  Matrix<float> b(a);                       // deep copy
  b.Permute(0, 2, 3, 1);                    // NCHW -> NHWC
  bad += !(b.GetDimLen(0) == 2 && b.GetDimLen(1) == 4 && b.GetDimLen(2) == 5 && b.GetDimLen(3) == 3);
  for (int n = 0; n < 2; ++n)
    for (int c = 0; c < 3; ++c)
      for (int h = 0; h < 4; ++h)
        for (int w = 0; w < 5; ++w) bad += !(b.GetEleAt(n, h, w, c) == a.GetEleAt(n, c, h, w));
  b.Permute(0, 3, 1, 2);                    // and back
  for (int i = 0; i < a.GetEleCnt(); ++i) bad += !(b.GetDataPtr()[i] == a.GetDataPtr()[i]);
  const float* before = a.GetDataPtr();
  a.Resize(6, 20);                          // same element count: relabel only, storage kept
  bad += !(a.GetDataPtr() == before && a.GetDimCnt() == 2 && a.GetDimStp(0) == 20 && a.GetEleAt(1, 3) == 23.0f);
  a.Resize(7, 3);                           // different count: re-created
  bad += !(a.GetEleCnt() == 21);
  Matrix<float> src(1, 4, 4, 6), dst(1, 4, 4, 3);
  for (int i = 0; i < src.GetEleCnt(); ++i) src.GetDataPtr()[i] = static_cast<float>(i + 1);
  src.GetSubMat(0, 0, 0, 3, &dst);          // channel group slice, as CalcFeatMap_ConvAprx uses it (:807)
  for (int h = 0; h < 4; ++h)
    for (int w = 0; w < 4; ++w)
      for (int c = 0; c < 3; ++c) bad += !(dst.GetEleAt(0, h, w, c) == src.GetEleAt(0, h, w, c + 3));
  Matrix<float> win(1, 3, 3, 6);
  src.GetSubMat(0, -1, 2, 0, &win);         // window hanging over the border: zero fill
  bad += !(win.GetEleAt(0, 0, 0, 0) == 0.0f && win.GetEleAt(0, 1, 0, 2) == src.GetEleAt(0, 0, 2, 2) &&
           win.GetEleAt(0, 2, 2, 0) == 0.0f);
  Matrix<uint8_t> e;
  Matrix<uint8_t> f;
  f = e;                                    // assignment of empty matrices must not crash
  Matrix<uint8_t> g(3, 2);
  g.SetEleAt(7, 2, 1);
  f = g;
  bad += !(f.GetEleAt(2, 1) == 7 && f.GetDataPtr() != g.GetDataPtr());
  Matrix<int> p2(2, 3);
  for (int i = 0; i < 6; ++i) p2.GetDataPtr()[i] = i;
  p2.Permute(1, 0);
  bad += !(p2.GetDimLen(0) == 3 && p2.GetEleAt(2, 1) == 5 && p2.GetEleAt(0, 1) == 3);
  return bad;
}

}  // extern "C"
