// bmp_img_io.cc — host pre-processing with the numerics of the reference's src/BmpImgIO.cc, so that a
// BMP produces the same network input bit for bit (tests/test_host_mirror.py checks this against the
// compiled reference): BGR planes :73-103, bilinear resize :105-178, centre crop :180-201, mean :203-224.
#include "../../include/BmpImgIO.h"

#include <fstream>
#include <vector>

#include "../../include/FileIO.h"

namespace {
const int kChn = 3;
const double kEps = 0.0000001;

uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }
}  // namespace

bool BmpImgIO::Init(const BmpImgIOPara& bmpImgIOPara) {
  para_ = bmpImgIOPara;
  return FileIO::ReadBinFile(para_.filePathMean, &mean_);
}

bool BmpImgIO::Load(const std::string& filePath, Matrix<float>* pImgDataFnal) {
  Matrix<float> raw, full;
  if (!decode(filePath, &raw)) return false;
  resize(raw, &full);
  if (para_.meanType == ENUM_MeanType::Full) {
    if (!subtractMean(&full)) return false;
    crop(full, pImgDataFnal);
    return true;
  }
  crop(full, pImgDataFnal);
  return subtractMean(pImgDataFnal);
}

// 14-byte file header + BITMAPINFOHEADER; pixel rows are padded to 4 bytes, bottom-up unless height < 0.
bool BmpImgIO::decode(const std::string& filePath, Matrix<float>* bgr) {
  std::ifstream f(filePath.c_str(), std::ios::binary);
  if (!f) {
    printf("[ERROR] cannot open the BMP image at %s\n", filePath.c_str());
    return false;
  }
  std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (buf.size() < 54 || buf[0] != 'B' || buf[1] != 'M') {
    printf("[ERROR] not a BMP file: %s\n", filePath.c_str());
    return false;
  }
  const uint32_t dataOff = rd32(&buf[10]);
  const int32_t wid = static_cast<int32_t>(rd32(&buf[18]));
  const int32_t heiRaw = static_cast<int32_t>(rd32(&buf[22]));
  const uint16_t bpp = rd16(&buf[28]);
  const uint32_t compression = rd32(&buf[30]);
  const int hei = heiRaw < 0 ? -heiRaw : heiRaw;
  if ((bpp != 24 && bpp != 32) || compression != 0 || wid <= 0 || hei <= 0) {
    printf("[ERROR] unsupported BMP flavour (%d bpp, compression %u): %s\n", bpp, compression, filePath.c_str());
    return false;
  }
  const int bytesPP = bpp / 8;
  const size_t stride = (static_cast<size_t>(wid) * bytesPP + 3) & ~static_cast<size_t>(3);
  if (buf.size() < dataOff + stride * hei) {
    printf("[ERROR] truncated BMP file: %s\n", filePath.c_str());
    return false;
  }
  bgr->Create(1, kChn, hei, wid);
  for (int y = 0; y < hei; ++y) {
    const int fileRow = heiRaw < 0 ? y : hei - 1 - y;
    const unsigned char* row = &buf[dataOff + stride * fileRow];
    for (int x = 0; x < wid; ++x)
      for (int c = 0; c < kChn; ++c) bgr->SetEleAt(row[x * bytesPP + c], 0, c, y, x);   // file order is B, G, R
  }
  return true;
}

void BmpImgIO::resize(const Matrix<float>& src, Matrix<float>* dst) const {
  const int hs = src.GetDimLen(2), ws = src.GetDimLen(3);
  float sh = static_cast<float>(hs - 1) / (para_.imgHeiFull - 1);
  float sw = static_cast<float>(ws - 1) / (para_.imgWidFull - 1);
  int hd = para_.imgHeiFull, wd = para_.imgWidFull;
  if (para_.reszType == ENUM_ReszType::Relaxed) {
    sh = std::min(sh, sw);
    sw = std::min(sh, sw);
    hd = static_cast<int>((hs - 1) / sh + kEps) + 1;
    wd = static_cast<int>((ws - 1) / sw + kEps) + 1;
  }
  printf("[INFO] resizing image from %d x %d to %d x %d\n", hs, ws, hd, wd);
  dst->Resize(1, kChn, hd, wd);
  for (int y = 0; y < hd; ++y) {
    const float yc = sh * y;
    const int y0 = std::max(0, static_cast<int>(yc));
    const int y1 = std::min(hs - 1, y0 + 1);
    const float wy0 = 1.0 - (yc - y0);
    const float wy1 = 1.0 - (y1 - yc);
    for (int x = 0; x < wd; ++x) {
      const float xc = sw * x;
      const int x0 = std::max(0, static_cast<int>(xc));
      const int x1 = std::min(ws - 1, x0 + 1);
      const float wx0 = 1.0 - (xc - x0);
      const float wx1 = 1.0 - (x1 - xc);
      for (int c = 0; c < kChn; ++c) {
        const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
        const float num = src.GetEleAt(0, c, y0, x0) * w00 + src.GetEleAt(0, c, y0, x1) * w01 +
                          src.GetEleAt(0, c, y1, x0) * w10 + src.GetEleAt(0, c, y1, x1) * w11;
        const float den = w00 + w01 + w10 + w11;
        dst->SetEleAt(num / den, 0, c, y, x);
      }
    }
  }
}

void BmpImgIO::crop(const Matrix<float>& src, Matrix<float>* dst) const {
  const int hs = src.GetDimLen(2), ws = src.GetDimLen(3);
  const int hd = para_.imgHeiCrop, wd = para_.imgWidCrop;
  const int oy = (hs - hd) / 2, ox = (ws - wd) / 2;
  dst->Resize(1, kChn, hd, wd);
  for (int c = 0; c < kChn; ++c)
    for (int y = 0; y < hd; ++y)
      memcpy(dst->GetDataPtr(0, c, y, 0), src.GetDataPtr(0, c, y + oy, ox), sizeof(float) * wd);
}

bool BmpImgIO::subtractMean(Matrix<float>* img) const {
  if (img->GetDimLen(2) != mean_.GetDimLen(1) || img->GetDimLen(3) != mean_.GetDimLen(2)) {
    printf("[ERROR] mismatch in the image size\n");
    printf("image %d x %d, mean %d x %d\n", img->GetDimLen(2), img->GetDimLen(3), mean_.GetDimLen(1), mean_.GetDimLen(2));
    return false;
  }
  const int n = img->GetEleCnt();
  float* p = img->GetDataPtr();
  const float* m = mean_.GetDataPtr();
  for (int i = 0; i < n; ++i) p[i] -= m[i];
  return true;
}
