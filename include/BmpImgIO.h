// BmpImgIO.h — BMP file -> network input, with the interface of the reference's BmpImgIO
// (include/BmpImgIO.h:18-83): Init(BmpImgIOPara) / Load(path, &img).  Host-side pre-processing only
// (decode 24-bit BMP to BGR planes, bilinear resize, mean subtraction, centre crop); it is outside the
// measured hot path.  The BMP decoder is this repository's own minimal reader
// (uncompressed 24/32-bit, bottom-up or top-down) instead of the reference's vendored bitmap_image.hpp.
#ifndef QCNN_HOST_BMPIMGIO_H_
#define QCNN_HOST_BMPIMGIO_H_

#include <string>

#include "../include/Common.h"
#include "../include/Matrix.h"

// Strict : resize to exactly H x W (aspect ratio may change)
// Relaxed: keep the aspect ratio, the smaller side becomes H (or W)
enum class ENUM_ReszType {Strict, Relaxed};
// Full: the mean image has the size of the resized image; Crop: the size of the cropped image
enum class ENUM_MeanType {Full, Crop};

typedef struct {
  ENUM_ReszType reszType;
  ENUM_MeanType meanType;
  int imgHeiFull;
  int imgWidFull;
  int imgHeiCrop;
  int imgWidCrop;
  std::string filePathMean;   // <C x H x W> fp32 .bin, BGR
} BmpImgIOPara;

class BmpImgIO {
 public:
  bool Init(const BmpImgIOPara& bmpImgIOPara);
  // out: [1, 3, imgHeiCrop, imgWidCrop] fp32, BGR, mean removed
  bool Load(const std::string& filePath, Matrix<float>* pImgDataFnal);

 private:
  BmpImgIOPara para_;
  Matrix<float> mean_;

  bool decode(const std::string& filePath, Matrix<float>* bgr);
  void resize(const Matrix<float>& src, Matrix<float>* dst) const;
  void crop(const Matrix<float>& src, Matrix<float>* dst) const;
  bool subtractMean(Matrix<float>* img) const;
};

#endif  // QCNN_HOST_BMPIMGIO_H_
