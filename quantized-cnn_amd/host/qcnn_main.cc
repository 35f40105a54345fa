// qcnn_main.cc — this repository's own command-line driver over the host mirror (the reference picks its
// mode by commenting lines in src/Main.cc:10-23; here it is an argument).
//
//   qcnn_main speed   [model] [data-root]          MODE 1: CaffeEva speed test over the evaluation subset
//   qcnn_main image   <file.bmp> [data-root]       MODE 2: one BMP -> top-5 (CaffeEvaWrapper)
//   qcnn_main convert <dir> <prefix> raw|compact   MODE 0: re-encode AlexNet assignment files (CvtAsmtEnc)
// Batch size / count / device / LUT builder come from QCNN_BATCH, QCNN_BATCHES, QCNN_DEVICE, QCNN_LUT.
#include <string>

#include "../../include/CaffeEva.h"
#include "../../include/CaffeEvaWrapper.h"
#include "../../include/CaffePara.h"
#include "../../include/StopWatch.h"

static int usage(void) {
  printf("usage: qcnn_main speed [model] [data-root] | image <file.bmp> [data-root] | convert <dir> <prefix> raw|compact\n");
  return 2;
}

int main(int argc, char* argv[]) {
  if (argc < 2) return usage();
  const std::string mode = argv[1];
  StopWatch sw;
  sw.Reset();
  sw.Resume();
  if (mode == "speed") {
    const std::string model = argc > 2 ? argv[2] : "AlexNet";
    const std::string root = argc > 3 ? argv[3] : ".";
    CaffeEva eva;
    eva.Init(true);
    eva.SetModelName(model);
    if (model == "VGG16") {
      eva.SetModelPath(root + "/VGG16/Bin.Files", "vgg16_aCaF");
      if (!eva.LoadDataset(root + "/ILSVRC12.224x224.PXL")) return 1;
    } else {
      eva.SetModelPath(root + "/AlexNet/Bin.Files", "bvlc_alexnet_aCaF");
      if (!eva.LoadDataset(root + "/ILSVRC12.227x227.IMG")) return 1;
    }
    if (!eva.LoadCaffePara()) return 1;
    eva.ExecForwardPass();
    eva.CalcPredAccu();
    eva.DispElpsTime();
  } else if (mode == "image") {
    if (argc < 3) return usage();
    const std::string root = argc > 3 ? argv[3] : ".";
    CaffeEvaWrapper w;
    CaffeEvaRslt r;
    r.clsCntPred = 5;
    if (!w.SetPath(root, root + "/Cls.Names/class_names.txt", root + "/Cls.Names/image_labels.txt") ||
        !w.SetModel(ENUM_CaffeModel::AlexNet, ENUM_CompMethod::Aprx) || !w.Proc(argv[2], &r)) {
      printf("[ERROR] %s\n", w.GetErrorMsg().c_str());
      return 1;
    }
    if (r.hasGrthClsName) printf("[INFO] Ground-truth class name: %s\n", r.clsNameGrth.c_str());
    for (int i = 0; i < r.clsCntPred; ++i)
      printf("[INFO] No. %d: %s (%d / %.4f)\n", i + 1, r.clsNameLst[i].c_str(), r.clsIdxLst[i], r.clsProbLst[i]);
  } else if (mode == "convert") {
    if (argc < 5) return usage();
    CaffePara para;
    para.Init(argv[2], argv[3]);
    para.ConfigLayer_AlexNet();
    const bool toCompact = std::string(argv[4]) == "compact";
    if (!para.CvtAsmtEnc(toCompact ? ENUM_AsmtEnc::Raw : ENUM_AsmtEnc::Compact,
                         toCompact ? ENUM_AsmtEnc::Compact : ENUM_AsmtEnc::Raw)) return 1;
  } else {
    return usage();
  }
  sw.Pause();
  printf("elapsed time: %.4f (s)\n", sw.GetTime());
  return 0;
}
