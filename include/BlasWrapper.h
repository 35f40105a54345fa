// BlasWrapper.h — placeholder kept so that code written against the reference's header set
// (include/BlasWrapper.h) still compiles.  The reference used it for cblas_saxpy (LUT build) and the
// vs*/sscal helpers (LRN) on the approximate path; here those run as HIP kernels
// (quantized-cnn_amd/csrc/qcnn_kernels.hip), so the host side needs no BLAS at all.
#ifndef QCNN_HOST_BLASWRAPPER_H_
#define QCNN_HOST_BLASWRAPPER_H_

typedef int CBLAS_INT;

#endif  // QCNN_HOST_BLASWRAPPER_H_
