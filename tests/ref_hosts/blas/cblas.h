/* TEST INFRASTRUCTURE -- the slice of the standard CBLAS interface that the reference's BLAS oracle calls
 * (include/Utility.h:14-16,76-103: cblas_sgemm / cblas_dgemm, row-major), so that the reference's own hosts can be built
 * with -DMM_HAS_BLAS as its CMake does when it finds a BLAS (CMakeLists.txt:75-85) on an image that ships a CBLAS library
 * (/opt/conda/lib/libmkl_rt.so, LP64: 32-bit integers) but no header for it.  Declarations of a public, vendor-neutral
 * interface (netlib's cblas.h); nothing here is an implementation.  Used only by tests/ref_hosts/build_ref_hosts.py. */
#ifndef MM_TEST_CBLAS_H
#define MM_TEST_CBLAS_H
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CblasRowMajor = 101, CblasColMajor = 102 } CBLAS_LAYOUT;
typedef CBLAS_LAYOUT CBLAS_ORDER;
typedef enum { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 } CBLAS_TRANSPOSE;

void cblas_sgemm(CBLAS_LAYOUT layout, CBLAS_TRANSPOSE trans_a, CBLAS_TRANSPOSE trans_b, int m, int n, int k, float alpha,
                 const float *a, int lda, const float *b, int ldb, float beta, float *c, int ldc);
void cblas_dgemm(CBLAS_LAYOUT layout, CBLAS_TRANSPOSE trans_a, CBLAS_TRANSPOSE trans_b, int m, int n, int k, double alpha,
                 const double *a, int lda, const double *b, int ldb, double beta, double *c, int ldc);

#ifdef __cplusplus
}
#endif
#endif
