// UnitTest.h — declaration of the three driver entry points that the reference's own src/UnitTest.cc
// defines and src/Main.cc calls (reference include/UnitTest.h:13-21).  This repository does not
// re-implement them: build/bin/QuanCNN_hip compiles byte-identical staged copies of those two files
// against the host mirror (quantized-cnn_amd/build.py: build_reference_driver).
#ifndef QCNN_HOST_UNITTEST_H_
#define QCNN_HOST_UNITTEST_H_

#include "../include/Common.h"

struct UnitTest {
  static void UT_CaffePara(void);        // parameter tooling: raw <-> compact assignment files
  static void UT_CaffeEva(void);         // MODE 1, speed test over the evaluation subset
  static void UT_CaffeEvaWrapper(void);  // MODE 2, one BMP image -> top-k
};

#endif  // QCNN_HOST_UNITTEST_H_
