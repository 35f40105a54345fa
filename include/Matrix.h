// Matrix.h — owning, row-major, 1..4-D host tensor with the public interface of the reference's
// Matrix<T> (include/Matrix.h:19-82), because that type sits on the API boundary:
// CaffePara::layerParaLst holds Matrix<> members and CaffeEva::ExecForwardPass(img, prob) takes them.
//
// Semantics preserved on purpose:
//   * Resize() only RELABELS the dimensions when the element count is unchanged (no reallocation,
//     contents kept) and re-creates the storage otherwise (reference :380-426);
//   * Permute(a, b, ...) physically reorders the data so that new dim i = old dim args[i];
//   * GetSubMat() zero-fills the destination and copies the overlapping window (reference :560-650).
// Fixed on purpose: the reference has a copy constructor but no assignment operator (double free on
// assignment) and a function-local static in the copy constructor; this class is a regular value type.
// This is host-side plumbing only — device tensors live behind include/qcnn_hip.h.
#ifndef QCNN_HOST_MATRIX_H_
#define QCNN_HOST_MATRIX_H_

#include <algorithm>

#include "../include/Common.h"

const int kMatDimCntMax = 4;

template <typename T>
class Matrix {
 public:
  Matrix(void) : rank_(0), buf_(nullptr) { clearDims(); }
  Matrix(const Matrix<T>& other) : rank_(0), buf_(nullptr) { clearDims(); assign(other); }
  explicit Matrix(const int m) : rank_(0), buf_(nullptr) { clearDims(); Create(m); }
  Matrix(const int m, const int n) : rank_(0), buf_(nullptr) { clearDims(); Create(m, n); }
  Matrix(const int m, const int n, const int p) : rank_(0), buf_(nullptr) { clearDims(); Create(m, n, p); }
  Matrix(const int m, const int n, const int p, const int q) : rank_(0), buf_(nullptr) {
    clearDims();
    Create(m, n, p, q);
  }
  Matrix(const int dimCnt, const int* dimLenLst) : rank_(0), buf_(nullptr) { clearDims(); Create(dimCnt, dimLenLst); }
  ~Matrix(void) { Destroy(); }
  Matrix<T>& operator=(const Matrix<T>& other) {
    if (this != &other) assign(other);
    return *this;
  }

  inline void Create(const int m) { const int d[1] = {m}; Create(1, d); }
  inline void Create(const int m, const int n) { const int d[2] = {m, n}; Create(2, d); }
  inline void Create(const int m, const int n, const int p) { const int d[3] = {m, n, p}; Create(3, d); }
  inline void Create(const int m, const int n, const int p, const int q) { const int d[4] = {m, n, p, q}; Create(4, d); }
  inline void Create(const int dimCnt, const int* dimLenLst) {
    if (dimCnt < 1 || dimCnt > kMatDimCntMax) {
      printf("[ERROR] invalid number of dimensions: %d\n", dimCnt);
      return;
    }
    Destroy();
    rank_ = dimCnt;
    for (int i = 0; i < dimCnt; ++i) len_[i] = dimLenLst[i];
    buf_ = new T[std::max(GetEleCnt(), 1)];
  }
  inline void Destroy(void) {
    delete[] buf_;
    buf_ = nullptr;
    rank_ = 0;
  }

  inline T* GetDataPtr(void) const { return buf_; }
  inline T* GetDataPtr(const int im) const { return buf_ + im; }
  inline T* GetDataPtr(const int im, const int in) const { return buf_ + offset(im, in); }
  inline T* GetDataPtr(const int im, const int in, const int ip) const { return buf_ + offset(im, in, ip); }
  inline T* GetDataPtr(const int im, const int in, const int ip, const int iq) const {
    return buf_ + offset(im, in, ip, iq);
  }

  inline int GetDimCnt(void) const { return rank_; }
  inline int GetDimLen(const int dimIdx) const {
    if (dimIdx < 0 || dimIdx >= kMatDimCntMax) {
      printf("[ERROR] invalid index of dimension: %d\n", dimIdx);
      return -1;
    }
    return len_[dimIdx];
  }
  // pointer step (in elements) of dimension dimIdx
  inline int GetDimStp(const int dimIdx) const {
    int s = 1;
    for (int i = rank_ - 1; i > dimIdx; --i) s *= len_[i];
    return s;
  }
  inline int GetEleCnt(void) const {
    if (rank_ == 0) return 0;
    int n = 1;
    for (int i = 0; i < rank_; ++i) n *= len_[i];
    return n;
  }
  inline void DispSizInfo(void) const {
    if (rank_ == 0) return;
    printf("[INFO] matrix size: %d", len_[0]);
    for (int i = 1; i < rank_; ++i) printf(" x %d", len_[i]);
    printf("\n");
  }

  inline void SetEleAt(const T val, const int im) { buf_[im] = val; }
  inline void SetEleAt(const T val, const int im, const int in) { buf_[offset(im, in)] = val; }
  inline void SetEleAt(const T val, const int im, const int in, const int ip) { buf_[offset(im, in, ip)] = val; }
  inline void SetEleAt(const T val, const int im, const int in, const int ip, const int iq) {
    buf_[offset(im, in, ip, iq)] = val;
  }
  inline T GetEleAt(const int im) const { return buf_[im]; }
  inline T GetEleAt(const int im, const int in) const { return buf_[offset(im, in)]; }
  inline T GetEleAt(const int im, const int in, const int ip) const { return buf_[offset(im, in, ip)]; }
  inline T GetEleAt(const int im, const int in, const int ip, const int iq) const { return buf_[offset(im, in, ip, iq)]; }

  inline void Resize(const int m) { const int d[1] = {m}; relabel(1, d); }
  inline void Resize(const int m, const int n) { const int d[2] = {m, n}; relabel(2, d); }
  inline void Resize(const int m, const int n, const int p) { const int d[3] = {m, n, p}; relabel(3, d); }
  inline void Resize(const int m, const int n, const int p, const int q) { const int d[4] = {m, n, p, q}; relabel(4, d); }

  void Permute(const int mSdx, const int nSdx) { const int o[2] = {mSdx, nSdx}; permute(2, o); }
  void Permute(const int mSdx, const int nSdx, const int pSdx) { const int o[3] = {mSdx, nSdx, pSdx}; permute(3, o); }
  void Permute(const int mSdx, const int nSdx, const int pSdx, const int qSdx) {
    const int o[4] = {mSdx, nSdx, pSdx, qSdx};
    permute(4, o);
  }

  void GetSubMat(const int imBeg, Matrix<T>* pMatDst) const { const int b[1] = {imBeg}; window(1, b, pMatDst); }
  void GetSubMat(const int imBeg, const int inBeg, Matrix<T>* pMatDst) const {
    const int b[2] = {imBeg, inBeg};
    window(2, b, pMatDst);
  }
  void GetSubMat(const int imBeg, const int inBeg, const int ipBeg, Matrix<T>* pMatDst) const {
    const int b[3] = {imBeg, inBeg, ipBeg};
    window(3, b, pMatDst);
  }
  void GetSubMat(const int imBeg, const int inBeg, const int ipBeg, const int iqBeg, Matrix<T>* pMatDst) const {
    const int b[4] = {imBeg, inBeg, ipBeg, iqBeg};
    window(4, b, pMatDst);
  }

 private:
  int rank_;
  int len_[kMatDimCntMax];
  T* buf_;

  void clearDims(void) { for (int i = 0; i < kMatDimCntMax; ++i) len_[i] = 0; }
  void assign(const Matrix<T>& o) {
    if (o.rank_ == 0) { Destroy(); return; }
    Create(o.rank_, o.len_);
    memcpy(buf_, o.buf_, sizeof(T) * GetEleCnt());
  }
  inline int offset(int a, int b) const { return a * len_[1] + b; }
  inline int offset(int a, int b, int c) const { return (a * len_[1] + b) * len_[2] + c; }
  inline int offset(int a, int b, int c, int d) const { return ((a * len_[1] + b) * len_[2] + c) * len_[3] + d; }

  void relabel(const int rank, const int* d) {
    int n = 1;
    for (int i = 0; i < rank; ++i) n *= d[i];
    if (GetEleCnt() != n) {
      Create(rank, d);
      return;
    }
    rank_ = rank;
    for (int i = 0; i < rank; ++i) len_[i] = d[i];
  }

  // new dimension i takes the role of old dimension order[i]
  void permute(const int rank, const int* order) {
    if (rank != rank_) {
      printf("[ERROR] Permute: %d indices for a %d-D matrix\n", rank, rank_);
      return;
    }
    const int n = GetEleCnt();
    int oldLen[kMatDimCntMax], oldStp[kMatDimCntMax], newLen[kMatDimCntMax], srcStp[kMatDimCntMax];
    for (int i = 0; i < rank; ++i) { oldLen[i] = len_[i]; oldStp[i] = GetDimStp(i); }
    for (int i = 0; i < kMatDimCntMax; ++i) { newLen[i] = 1; srcStp[i] = 0; }
    for (int i = 0; i < rank; ++i) { newLen[i] = oldLen[order[i]]; srcStp[i] = oldStp[order[i]]; }
    T* fresh = new T[std::max(n, 1)];
    T* w = fresh;
    for (int a = 0; a < newLen[0]; ++a)
      for (int b = 0; b < newLen[1]; ++b)
        for (int c = 0; c < newLen[2]; ++c) {
          const T* r = buf_ + a * srcStp[0] + b * srcStp[1] + c * srcStp[2];
          for (int d = 0; d < newLen[3]; ++d, r += srcStp[3]) *w++ = *r;
        }
    delete[] buf_;
    buf_ = fresh;
    for (int i = 0; i < rank; ++i) len_[i] = newLen[i];
  }

  // copy the part of *this that overlaps [beg, beg + dst.len) into dst, zero elsewhere
  void window(const int rank, const int* beg, Matrix<T>* dst) const {
    memset(dst->buf_, 0, sizeof(T) * dst->GetEleCnt());
    int lo[kMatDimCntMax], hi[kMatDimCntMax], b4[kMatDimCntMax];
    for (int i = 0; i < kMatDimCntMax; ++i) { lo[i] = 0; hi[i] = 0; b4[i] = 0; }
    for (int i = 0; i < rank; ++i) {
      b4[i] = beg[i];
      lo[i] = std::max(0, -beg[i]);
      hi[i] = std::min(dst->len_[i] - 1, len_[i] - 1 - beg[i]);
      if (hi[i] < lo[i]) return;
    }
    const int last = rank - 1;
    const int run = hi[last] - lo[last] + 1;
    int idx[kMatDimCntMax] = {lo[0], lo[1], lo[2], lo[3]};
    for (;;) {
      int so = 0, doff = 0;
      for (int i = 0; i < rank; ++i) {
        const int di = (i == last) ? lo[last] : idx[i];
        so = so * len_[i] + di + b4[i];
        doff = doff * dst->len_[i] + di;
      }
      memcpy(dst->buf_ + doff, buf_ + so, sizeof(T) * run);
      int k = last - 1;
      while (k >= 0) {
        if (++idx[k] <= hi[k]) break;
        idx[k] = lo[k];
        --k;
      }
      if (k < 0) break;
    }
  }
};

#endif  // QCNN_HOST_MATRIX_H_
