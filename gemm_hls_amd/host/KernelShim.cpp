// Build-time configured kernel library: the role of the reference's `mmkernel` target
// (CMakeLists.txt:138-146, kernel/Top.cpp compiled for ONE (MM_DATA_TYPE, MM_MAP_OP, MM_REDUCE_OP,
// MM_TRANSPOSED_A, MM_DYNAMIC_SIZES | MM_SIZE_*) choice and linked by TestSimulation / RunHardware).
// It exports the reference's symbol with the reference's exact signature for that choice
// (include/MatrixMultiplication.h:155-171):
//
//   MM_DYNAMIC_SIZES            void MatrixMultiplicationKernel(a, b, c, size_n, size_k, size_m)
//   static sizes (MM_SIZE_*)    void MatrixMultiplicationKernel(a, b, c)          <- 3-pointer form
//   MM_TRANSPOSED_A             `a` is the K x N array (MemoryPackN_t const a[] in the reference)
//
// and forwards to the runtime-dispatched device library (mm_gemm_host, include/mm_gemm.h), so the
// reference's test/TestSimulation.cpp:66-69 call compiles and links against this file unchanged.
// DataPack arrays are layout-compatible with plain Data_t arrays (include/Utility.h:44-63), hence
// the element-typed pointers.  Link this library BEFORE libmm_gemm_amd.so: the latter also exports a
// run-time configured 6-argument MatrixMultiplicationKernel (mm_set_default_config); the first
// definition in link order is the one a caller binds to.
#include <cstdio>
#include <cstdlib>

#define MM_GEMM_NO_KERNEL_SYMBOL
#include "HostConfig.h"

// -DMM_DEFAULT_PATH=2 builds the shim over MM_PATH_SPLIT (float (Multiply, Add) only; include/mm_gemm.h)
#ifndef MM_DEFAULT_PATH
#define MM_DEFAULT_PATH MM_PATH_AUTO
#endif

namespace {
void Run(const void *a, const void *b, void *c, unsigned size_n, unsigned size_k, unsigned size_m) {
  static const char *const pinned = mmhost::ApplyBuildTimeTile();   // once: -DMM_MEMORY_TILE_SIZE_N/M
  (void)pinned;
  const mm_config_t cfg = {mmhost::DTypeOf<Data_t>::value, OperatorMap::code, OperatorReduce::code, (mm_path_t)MM_DEFAULT_PATH,
                           kTransposedA ? MM_A_TRANSPOSED : MM_A_ROW_MAJOR};
  if (mm_gemm_host(&cfg, a, b, c, size_n, size_k, size_m) != MM_OK) {
    std::fprintf(stderr, "MatrixMultiplicationKernel failed: %s\n", mm_last_error());
    std::abort();  // the reference's symbol returns void; failing silently would fake a result
  }
}
}  // namespace

extern "C" {
#ifdef MM_DYNAMIC_SIZES
__attribute__((visibility("default"))) void MatrixMultiplicationKernel(Data_t const a[], Data_t const b[], Data_t c[],
                                                                       const unsigned size_n, const unsigned size_k,
                                                                       const unsigned size_m) {
  Run(a, b, c, size_n, size_k, size_m);
}
#else
__attribute__((visibility("default"))) void MatrixMultiplicationKernel(Data_t const a[], Data_t const b[], Data_t c[]) {
  Run(a, b, c, kSizeN, kSizeK, kSizeM);
}
// the sizes this library was built for, so that a caller (or a test) can check them
__attribute__((visibility("default"))) void MatrixMultiplicationKernelSizes(unsigned *n, unsigned *k, unsigned *m) {
  *n = kSizeN;
  *k = kSizeK;
  *m = kSizeM;
}
#endif
}
