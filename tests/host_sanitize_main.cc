// Driver of tests/test_host_sanitizers.py: the device-free part of the C++ host mirror (Matrix, FileIO incl. the
// .cbn codec, CaffePara's loaders and encoders, BmpImgIO) built with -fsanitize=address,undefined.  The reference
// itself has no sanitizer coverage and several latent defects in exactly these classes: uninitialised
// pointers, a missing Matrix::operator=, function-local static scratch.
#include <cstdio>
#include <cstdlib>
extern "C" {
int qh_matrix_selftest(void);
int qh_para_load(const char* model, const char* dir, const char* pfx, int raw, int* layerCnt, int* dims, double* sums);
int qh_para_convert(const char* dir, const char* pfx, int toCompact);
int qh_cbn_rewrite(const char* inPath, const char* outPath, int bits);
int qh_bmp_load(const char* meanPath, const char* bmpPath, int full, int crop, int relaxed, float* out);
}
int main(int argc, char** argv) {
  if (argc < 3) return 64;
  const char* dir = argv[1];
  const char* pfx = argv[2];
  int bad = qh_matrix_selftest();
  if (bad) { printf("matrix selftest: %d failures\n", bad); return 1; }
  static int dims[64 * 8];
  static double sums[64 * 3];
  int n = 0;
  if (qh_para_load("AlexNet", dir, pfx, 0, &n, dims, sums)) { printf("LoadLayerPara(Compact) failed\n"); return 2; }
  if (qh_para_convert(dir, pfx, 0)) { printf("CvtAsmtEnc Compact -> Raw failed\n"); return 3; }
  if (qh_para_load("AlexNet", dir, pfx, 1, &n, dims, sums)) { printf("LoadLayerPara(Raw) failed\n"); return 4; }
  if (qh_para_convert(dir, pfx, 1)) { printf("CvtAsmtEnc Raw -> Compact failed\n"); return 5; }
  if (qh_para_load("NoSuchModel", dir, pfx, 0, &n, dims, sums) == 0) return 6;          // error paths must stay clean too
  if (qh_para_load("AlexNet", "/nonexistent", pfx, 0, &n, dims, sums) == 0) return 7;
  if (argc >= 5) {                                                                     // mean image + BMP (where staged)
    static float img[3 * 227 * 227];
    if (qh_bmp_load(argv[3], argv[4], 256, 227, 0, img)) { printf("BmpImgIO::Load failed\n"); return 8; }
    if (qh_bmp_load(argv[3], "/nonexistent.bmp", 256, 227, 0, img) == 0) return 9;
  }
  printf("host mirror under ASan/UBSan: OK (%d layers)\n", n);
  return 0;
}
