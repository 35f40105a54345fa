// FileIO.h — readers/writers of the three on-disk formats of the Quantized-CNN parameter and dataset
// files, with the interface of the reference's FileIO (include/FileIO.h:23-52).
//
//   .bin  int32 dimCnt; int32 dims[dimCnt]; raw little-endian T[] (row-major)      reference :55-107, :240-297
//   .cbn  int32 dimCnt; int32 dims[]; int32 bitCntPerEle; then 4096-byte blocks, each packing
//         floor(32768 / bits) values MSB-first; values never straddle a block      reference :109-178, :299-350
//   .txt  "dimCnt d0 d1 ..." then the values                                       reference :180-238, :352-395
//
// Contract kept bit-for-bit: a .cbn file stores (index - 1); ReadCbnFile hands back 1-based indices
// (reference :165) and WriteCbnFile expects 1-based input (reference :329-335) — CaffePara::LoadLayerPara
// is the one that converts to 0-based.  The implementation is this repository's own (whole-payload
// reads, a bit cursor instead of the byte-state machine).
#ifndef QCNN_HOST_FILEIO_H_
#define QCNN_HOST_FILEIO_H_

#include <algorithm>
#include <string>
#include <vector>

#include "../include/Common.h"
#include "../include/Matrix.h"

class FileIO {
 public:
  template <typename T> static bool ReadBinFile(const std::string& filePath, Matrix<T>* pDataLst);
  template <typename T> static bool ReadCbnFile(const std::string& filePath, Matrix<T>* pDataLst);
  template <typename T> static bool ReadTxtFile(const std::string& filePath, Matrix<T>* pDataLst);
  template <typename T> static bool WriteBinFile(const std::string& filePath, const Matrix<T>& dataLst);
  template <typename T>
  static bool WriteCbnFile(const std::string& filePath, const Matrix<T>& dataLst, const int bitCntPerEle);
  template <typename T> static bool WriteTxtFile(const std::string& filePath, const Matrix<T>& dataLst);

 private:
  static const int kBlockBytes = 4096;
  static bool readDims(FILE* f, std::vector<int32_t>* dims) {
    int32_t rank = 0;
    if (fread(&rank, sizeof(int32_t), 1, f) != 1 || rank < 1 || rank > kMatDimCntMax) return false;
    dims->resize(rank);
    return fread(dims->data(), sizeof(int32_t), rank, f) == static_cast<size_t>(rank);
  }
  template <typename T> static void writeDims(FILE* f, const Matrix<T>& m) {
    const int32_t rank = m.GetDimCnt();
    fwrite(&rank, sizeof(int32_t), 1, f);
    for (int i = 0; i < rank; ++i) {
      const int32_t len = m.GetDimLen(i);
      fwrite(&len, sizeof(int32_t), 1, f);
    }
  }
  template <typename T> static const char* scanFmt(void);
  template <typename T> static const char* printFmt(void);
};

template <typename T>
bool FileIO::ReadBinFile(const std::string& filePath, Matrix<T>* pDataLst) {
  FILE* f = fopen(filePath.c_str(), "rb");
  if (f == nullptr) {
    printf("[ERROR] could not open file at %s\n", filePath.c_str());
    return false;
  }
  std::vector<int32_t> dims;
  bool ok = readDims(f, &dims);
  if (ok) {
    pDataLst->Create(static_cast<int>(dims.size()), dims.data());
    const size_t n = pDataLst->GetEleCnt();
    ok = fread(pDataLst->GetDataPtr(), sizeof(T), n, f) == n;
  }
  fclose(f);
  if (!ok) printf("[ERROR] malformed or truncated .bin file: %s\n", filePath.c_str());
  return ok;
}

template <typename T>
bool FileIO::ReadCbnFile(const std::string& filePath, Matrix<T>* pDataLst) {
  FILE* f = fopen(filePath.c_str(), "rb");
  if (f == nullptr) {
    printf("[ERROR] could not open file at %s\n", filePath.c_str());
    return false;
  }
  std::vector<int32_t> dims;
  int32_t bits = 0;
  bool ok = readDims(f, &dims) && fread(&bits, sizeof(int32_t), 1, f) == 1 && bits >= 1 && bits <= 16;
  if (ok) {
    pDataLst->Create(static_cast<int>(dims.size()), dims.data());
    const int n = pDataLst->GetEleCnt();
    const int perBlock = kBlockBytes * 8 / bits;
    const int blocks = (n + perBlock - 1) / perBlock;
    std::vector<uint8_t> raw(static_cast<size_t>(blocks) * kBlockBytes);
    ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
    T* out = pDataLst->GetDataPtr();
    for (int i = 0; ok && i < n; ++i) {
      const uint8_t* blk = raw.data() + static_cast<size_t>(i / perBlock) * kBlockBytes;
      const int bit0 = (i % perBlock) * bits;
      unsigned v = 0;
      for (int t = 0; t < bits; ++t) {
        const int bit = bit0 + t;
        v = (v << 1) | ((blk[bit >> 3] >> (7 - (bit & 7))) & 1u);
      }
      out[i] = static_cast<T>(v + 1);   // stored value is index - 1
    }
  }
  fclose(f);
  if (!ok) printf("[ERROR] malformed or truncated .cbn file: %s\n", filePath.c_str());
  return ok;
}

template <typename T>
bool FileIO::WriteBinFile(const std::string& filePath, const Matrix<T>& dataLst) {
  FILE* f = fopen(filePath.c_str(), "wb");
  if (f == nullptr) {
    printf("[ERROR] could not open file at %s\n", filePath.c_str());
    return false;
  }
  writeDims(f, dataLst);
  const size_t n = dataLst.GetEleCnt();
  const bool ok = fwrite(dataLst.GetDataPtr(), sizeof(T), n, f) == n;
  fclose(f);
  return ok;
}

template <typename T>
bool FileIO::WriteCbnFile(const std::string& filePath, const Matrix<T>& dataLst, const int bitCntPerEle) {
  if (bitCntPerEle < 1 || bitCntPerEle > 16) {
    printf("[ERROR] invalid bitCntPerEle: %d\n", bitCntPerEle);
    return false;
  }
  FILE* f = fopen(filePath.c_str(), "wb");
  if (f == nullptr) {
    printf("[ERROR] could not open file at %s\n", filePath.c_str());
    return false;
  }
  writeDims(f, dataLst);
  const int32_t bits = bitCntPerEle;
  fwrite(&bits, sizeof(int32_t), 1, f);
  const int n = dataLst.GetEleCnt();
  const int perBlock = kBlockBytes * 8 / bits;
  const int blocks = (n + perBlock - 1) / perBlock;
  const T* in = dataLst.GetDataPtr();
  std::vector<uint8_t> blk(kBlockBytes);
  bool ok = true;
  for (int b = 0; ok && b < blocks; ++b) {
    std::fill(blk.begin(), blk.end(), 0);
    const int cnt = std::min(perBlock, n - b * perBlock);
    for (int j = 0; j < cnt; ++j) {
      const unsigned v = static_cast<unsigned>(in[b * perBlock + j]) - 1u;   // 1-based in, index - 1 stored
      for (int t = 0; t < bits; ++t) {
        const int bit = j * bits + t;
        if ((v >> (bits - 1 - t)) & 1u) blk[bit >> 3] |= static_cast<uint8_t>(1u << (7 - (bit & 7)));
      }
    }
    ok = fwrite(blk.data(), 1, kBlockBytes, f) == static_cast<size_t>(kBlockBytes);
  }
  fclose(f);
  return ok;
}

template <> inline const char* FileIO::scanFmt<uint8_t>(void) { return "%hhu"; }
template <> inline const char* FileIO::scanFmt<int8_t>(void) { return "%hhd"; }
template <> inline const char* FileIO::scanFmt<uint16_t>(void) { return "%hu"; }
template <> inline const char* FileIO::scanFmt<int16_t>(void) { return "%hd"; }
template <> inline const char* FileIO::scanFmt<uint32_t>(void) { return "%u"; }
template <> inline const char* FileIO::scanFmt<int32_t>(void) { return "%d"; }
template <> inline const char* FileIO::scanFmt<float>(void) { return "%f"; }
template <> inline const char* FileIO::scanFmt<double>(void) { return "%lf"; }
template <> inline const char* FileIO::printFmt<uint8_t>(void) { return "%hhu"; }
template <> inline const char* FileIO::printFmt<int8_t>(void) { return "%hhd"; }
template <> inline const char* FileIO::printFmt<uint16_t>(void) { return "%hu"; }
template <> inline const char* FileIO::printFmt<int16_t>(void) { return "%hd"; }
template <> inline const char* FileIO::printFmt<uint32_t>(void) { return "%u"; }
template <> inline const char* FileIO::printFmt<int32_t>(void) { return "%d"; }
template <> inline const char* FileIO::printFmt<float>(void) { return "%.4f"; }
template <> inline const char* FileIO::printFmt<double>(void) { return "%.4f"; }

template <typename T>
bool FileIO::ReadTxtFile(const std::string& filePath, Matrix<T>* pDataLst) {
  FILE* f = fopen(filePath.c_str(), "r");
  if (f == nullptr) {
    printf("[ERROR] could not open file at %s\n", filePath.c_str());
    return false;
  }
  int rank = 0;
  bool ok = fscanf(f, "%d", &rank) == 1 && rank >= 1 && rank <= kMatDimCntMax;
  int dims[kMatDimCntMax] = {0, 0, 0, 0};
  for (int i = 0; ok && i < rank; ++i) ok = fscanf(f, "%d", &dims[i]) == 1;
  if (ok) {
    pDataLst->Create(rank, dims);
    T* p = pDataLst->GetDataPtr();
    const int n = pDataLst->GetEleCnt();
    for (int i = 0; ok && i < n; ++i) ok = fscanf(f, scanFmt<T>(), p + i) == 1;
  }
  fclose(f);
  if (!ok) printf("[ERROR] malformed .txt matrix file: %s\n", filePath.c_str());
  return ok;
}

template <typename T>
bool FileIO::WriteTxtFile(const std::string& filePath, const Matrix<T>& dataLst) {
  FILE* f = fopen(filePath.c_str(), "w");
  if (f == nullptr) {
    printf("[ERROR] could not open file at %s\n", filePath.c_str());
    return false;
  }
  const int rank = dataLst.GetDimCnt();
  fprintf(f, "%d", rank);
  for (int i = 0; i < rank; ++i) fprintf(f, " %d", dataLst.GetDimLen(i));
  fprintf(f, "\n");
  const int n = dataLst.GetEleCnt();
  const int rowLen = dataLst.GetDimLen(rank - 1);
  const T* p = dataLst.GetDataPtr();
  for (int i = 0; i < n; ++i) {
    fprintf(f, printFmt<T>(), p[i]);
    fputc(((i + 1) % rowLen) ? ' ' : '\n', f);
  }
  fclose(f);
  return true;
}

#endif  // QCNN_HOST_FILEIO_H_
