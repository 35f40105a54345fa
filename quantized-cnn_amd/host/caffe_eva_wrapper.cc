// caffe_eva_wrapper.cc — host façade, mirrors the reference's src/CaffeEvaWrapper.cc (SetPath :15-42,
// SetModel :44-151, Proc :153-209, label files :219-283).
#include "../../include/CaffeEvaWrapper.h"

#include <fstream>
#include <sstream>

namespace {
struct ModelEntry {
  ENUM_CaffeModel id;
  const char* name;
  const char* dir;
  const char* prefix;
  bool relaxedResize;   // VggCnnS: keep the aspect ratio and use a crop-sized mean image
  int crop;
};
const ModelEntry kModels[] = {
    {ENUM_CaffeModel::AlexNet, "AlexNet", "AlexNet", "bvlc_alexnet_aCaF", false, 227},
    {ENUM_CaffeModel::CaffeNet, "CaffeNet", "CaffeNet", "bvlc_caffenet_aCaF", false, 227},
    {ENUM_CaffeModel::VggCnnS, "VggCnnS", "VggCnnS", "vgg_cnn_s_aCaF", true, 224},
    {ENUM_CaffeModel::CaffeNetFGB, "CaffeNetFGB", "CaffeNetFGB", "bvlc_caffenetfgb_aCaF", false, 227},
    {ENUM_CaffeModel::CaffeNetFGD, "CaffeNetFGD", "CaffeNetFGD", "bvlc_caffenetfgd_aCaF", false, 227},
};
}  // namespace

CaffeEvaWrapper::CaffeEvaWrapper(void) : caffeModel(ENUM_CaffeModel::AlexNet), compMethod(ENUM_CompMethod::Aprx) {
  ClrErrorMsg();
}

std::string CaffeEvaWrapper::GetErrorMsg(void) { return errorMsg; }
void CaffeEvaWrapper::ClrErrorMsg(void) { errorMsg = ""; }

// file name without directories and without its extension(s)
std::string CaffeEvaWrapper::baseName(const std::string& filePath) {
  const size_t slash = filePath.find_last_of('/');
  std::string name = (slash == std::string::npos) ? filePath : filePath.substr(slash + 1);
  const size_t dot = name.find('.');
  return (dot == std::string::npos) ? name : name.substr(0, dot);
}

bool CaffeEvaWrapper::SetPath(const std::string& mainDirPathSrc, const std::string& clsNameFilePath,
                              const std::string& imgLablFilePath) {
  mainDirPath = mainDirPathSrc;
  std::ifstream names(clsNameFilePath.c_str());
  if (!names) {
    errorMsg = "[CaffeEvaWrapper::SetPath] could not open file: " + clsNameFilePath;
    return false;
  }
  clsNameLst.clear();
  for (std::string line; std::getline(names, line);) clsNameLst.push_back(line);
  if (imgLablFilePath.empty()) return true;
  std::ifstream labels(imgLablFilePath.c_str());
  if (!labels) {
    errorMsg = "[CaffeEvaWrapper::SetPath] could not open file: " + imgLablFilePath;
    return false;
  }
  clsNameGrthLst.clear();
  std::string file;
  int cls;
  while (labels >> file >> cls) {
    ClsNameGrthStr e;
    e.fileName = baseName(file);
    e.clsNameGrth = (cls >= 0 && cls < static_cast<int>(clsNameLst.size())) ? clsNameLst[cls] : std::string("?");
    clsNameGrthLst.push_back(e);
  }
  return true;
}

bool CaffeEvaWrapper::SetModel(const ENUM_CaffeModel& caffeModelSrc, const ENUM_CompMethod& compMethodSrc) {
  caffeModel = caffeModelSrc;
  compMethod = compMethodSrc;
  if (caffeModel == ENUM_CaffeModel::VGG16) {   // as the reference: not wired into the wrapper (src/CaffeEvaWrapper.cc:77-80)
    printf("[FATAL ERROR] VGG-16 is not supported (for now)\n");
    errorMsg = "[CaffeEvaWrapper::SetModel] unsupported caffe model name";
    return false;
  }
  const ModelEntry* m = nullptr;
  for (size_t i = 0; i < sizeof(kModels) / sizeof(kModels[0]); ++i)
    if (kModels[i].id == caffeModel) m = &kModels[i];
  if (m == nullptr) {
    printf("[FATAL ERROR] unrecognized <ENUM_CaffeModel> value\n");
    errorMsg = "[CaffeEvaWrapper::SetModel] unrecognized caffe model name";
    return false;
  }
  bmpImgIOPara.reszType = m->relaxedResize ? ENUM_ReszType::Relaxed : ENUM_ReszType::Strict;
  bmpImgIOPara.meanType = m->relaxedResize ? ENUM_MeanType::Crop : ENUM_MeanType::Full;
  bmpImgIOPara.imgHeiFull = 256;
  bmpImgIOPara.imgWidFull = 256;
  bmpImgIOPara.imgHeiCrop = m->crop;
  bmpImgIOPara.imgWidCrop = m->crop;
  bmpImgIOPara.filePathMean = mainDirPath + "/" + m->dir + "/imagenet_mean.single.bin";
  if (!bmpImgIOObj.Init(bmpImgIOPara)) {
    errorMsg = "[CaffeEvaWrapper::SetModel] could not open the mean image file";
    return false;
  }
  caffeEvaObj.Init(compMethodSrc == ENUM_CompMethod::Aprx);
  caffeEvaObj.SetModelName(m->name);
  caffeEvaObj.SetModelPath(mainDirPath + "/" + m->dir + "/Bin.Files", m->prefix);
  if (!caffeEvaObj.LoadCaffePara()) {
    errorMsg = "[CaffeEvaWrapper::SetModel] could not load model files";
    return false;
  }
  return true;
}

bool CaffeEvaWrapper::Proc(const std::string& filePathProcImg, CaffeEvaRslt* pCaffeEvaRslt) {
  Matrix<float> img;
  if (!bmpImgIOObj.Load(filePathProcImg, &img)) {
    errorMsg = "[CaffeEvaWrapper::Proc] could open the BMP file";
    return false;
  }
  Matrix<float> prob;
  caffeEvaObj.ExecForwardPass(img, &prob);
  if (!caffeEvaObj.GetErrorMsg().empty()) {              // the forward pass failed: do not rank a zero vector
    errorMsg = "[CaffeEvaWrapper::Proc] forward pass failed: " + caffeEvaObj.GetErrorMsg();
    return false;
  }
  pCaffeEvaRslt->timeTotal = caffeEvaObj.DispElpsTime();

  const std::string key = baseName(filePathProcImg);
  pCaffeEvaRslt->hasGrthClsName = false;
  for (size_t i = 0; i < clsNameGrthLst.size(); ++i) {
    if (clsNameGrthLst[i].fileName == key) {
      pCaffeEvaRslt->hasGrthClsName = true;
      pCaffeEvaRslt->clsNameGrth = clsNameGrthLst[i].clsNameGrth;
      break;
    }
  }
  // k rounds of arg-max with the winner zeroed (src/CaffeEvaWrapper.cc:188-206)
  const int n = prob.GetEleCnt();
  float* p = prob.GetDataPtr();
  pCaffeEvaRslt->clsIdxLst.clear();
  pCaffeEvaRslt->clsProbLst.clear();
  pCaffeEvaRslt->clsNameLst.clear();
  for (int r = 0; r < pCaffeEvaRslt->clsCntPred && n > 0; ++r) {
    int best = 0;
    for (int c = 1; c < n; ++c)
      if (p[best] < p[c]) best = c;
    pCaffeEvaRslt->clsIdxLst.push_back(best);
    pCaffeEvaRslt->clsProbLst.push_back(p[best]);
    pCaffeEvaRslt->clsNameLst.push_back(best < static_cast<int>(clsNameLst.size()) ? clsNameLst[best] : std::string("?"));
    p[best] = 0.0f;
  }
  return true;
}
