// CaffeEvaWrapper.h — "one BMP file in, top-k class names out" façade with the interface of the
// reference's CaffeEvaWrapper (include/CaffeEvaWrapper.h:18-107): SetPath / SetModel / Proc /
// GetErrorMsg / ClrErrorMsg, the CaffeEvaRslt result record, the two model/method enums.  Same error
// convention: bool + a message kept in the object (src/CaffeEvaWrapper.cc:211-217).
#ifndef QCNN_HOST_CAFFEEVAWRAPPER_H_
#define QCNN_HOST_CAFFEEVAWRAPPER_H_

#include <string>
#include <vector>

#include "../include/Common.h"
#include "../include/BmpImgIO.h"
#include "../include/CaffeEva.h"

enum class ENUM_CaffeModel {AlexNet, CaffeNet, VggCnnS, VGG16, CaffeNetFGB, CaffeNetFGD};
enum class ENUM_CompMethod {Prec, Aprx};

typedef struct {
  int clsCntPred;                        // in: how many labels to return
  float timeTotal;                       // out: device time of the forward pass, seconds
  bool hasGrthClsName;
  std::string clsNameGrth;
  std::vector<int> clsIdxLst;
  std::vector<float> clsProbLst;
  std::vector<std::string> clsNameLst;
} CaffeEvaRslt;

typedef struct {
  std::string fileName;
  std::string clsNameGrth;
} ClsNameGrthStr;
typedef std::vector<ClsNameGrthStr> ClsNameGrthLst;

class CaffeEvaWrapper {
 public:
  CaffeEvaWrapper(void);
  // mainDirPathSrc: data root (model parameters, mean images); clsNameFilePath: one class name per line;
  // imgLablFilePath (optional): "<file> <class index>" per line for ground truth
  bool SetPath(const std::string& mainDirPathSrc, const std::string& clsNameFilePath,
               const std::string& imgLablFilePath = "");
  bool SetModel(const ENUM_CaffeModel& caffeModelSrc, const ENUM_CompMethod& compMethodSrc);
  bool Proc(const std::string& filePathProcImg, CaffeEvaRslt* pCaffeEvaRslt);
  std::string GetErrorMsg(void);
  void ClrErrorMsg(void);

 private:
  std::string mainDirPath;
  ENUM_CaffeModel caffeModel;
  ENUM_CompMethod compMethod;
  BmpImgIOPara bmpImgIOPara;
  BmpImgIO bmpImgIOObj;
  CaffeEva caffeEvaObj;
  std::vector<std::string> clsNameLst;
  ClsNameGrthLst clsNameGrthLst;
  std::string errorMsg;

  static std::string baseName(const std::string& filePath);
};

#endif  // QCNN_HOST_CAFFEEVAWRAPPER_H_
