// TestSimulation -- the role of the reference's CTest binary (test/TestSimulation.cpp) with the GPU
// behind the symbol it calls.  Command line, size checks and messages are the reference's:
//     TestSimulation N K M        -> "Running simulation...", "Verifying results...",
//                                    "Matrix-matrix multiplication successfully verified."
// The single call MatrixMultiplicationKernel(a, b, c[, N, K, M]) takes HOST pointers; in the
// reference it runs the HLS kernel as host threads (the `mmkernel` library, CMakeLists.txt:138-146),
// here it binds to this build's kernel shim (host/KernelShim.cpp, same configuration macros, same
// signature incl. the 3-pointer static-size form) and runs on MI355X device 0.
#include <iostream>
#include <random>
#include <string>
#include <vector>

#define MM_GEMM_NO_KERNEL_SYMBOL  // the typed declaration below is the reference's (MatrixMultiplication.h:155-171)
#include "HostReference.h"

extern "C" {
#ifdef MM_DYNAMIC_SIZES
void MatrixMultiplicationKernel(Data_t const a[], Data_t const b[], Data_t c[], const unsigned size_n,
                                const unsigned size_k, const unsigned size_m);
#else
void MatrixMultiplicationKernel(Data_t const a[], Data_t const b[], Data_t c[]);
#endif
}

namespace {

struct Sizes {
  unsigned n, k, m;
};

// Returns false (after printing the reference's message) when the arguments are unusable.
bool ParseSizes(int argc, char **argv, Sizes &out) {
#ifdef MM_DYNAMIC_SIZES
  if (argc != 4) {
    std::cerr << "Usage: ./TestSimulation N K M" << std::endl;
    return false;
  }
  out = {static_cast<unsigned>(std::stoul(argv[1])), static_cast<unsigned>(std::stoul(argv[2])),
         static_cast<unsigned>(std::stoul(argv[3]))};
  const char *bad = nullptr;
  if (out.k % kMemoryWidthK != 0) bad = "K";
  else if (out.m % kMemoryWidthM != 0) bad = "M";
  if (bad) {
    std::cerr << bad << " must be divisable by memory width." << std::endl;
    return false;
  }
#else
  (void)argc;
  (void)argv;
  out = {static_cast<unsigned>(kSizeN), static_cast<unsigned>(kSizeK), static_cast<unsigned>(kSizeM)};
#endif
  return true;
}

// Seed kSeed, uniform on [1, 10] (integer or real by element type), A entirely before B.
void SeededInputs(std::vector<Data_t> &a, std::vector<Data_t> &b) {
  std::default_random_engine engine(kSeed);
  using Uniform = std::conditional<std::is_integral<Data_t>::value, std::uniform_int_distribution<unsigned long>,
                                   std::uniform_real_distribution<double>>::type;
  Uniform one_to_ten(1, 10);
  for (std::vector<Data_t> *matrix : {&a, &b})
    for (Data_t &element : *matrix) element = Data_t(one_to_ten(engine));
}

}  // namespace

int main(int argc, char **argv) {
  Sizes sz;
  if (!ParseSizes(argc, argv, sz)) return 1;

  std::vector<Data_t> a(static_cast<size_t>(sz.n) * sz.k), b(static_cast<size_t>(sz.k) * sz.m);
  SeededInputs(a, b);
  std::vector<Data_t> expected(static_cast<size_t>(sz.n) * sz.m, Data_t(0)), computed(expected.size(), Data_t(0));
  // MM_TRANSPOSED_A: the N*K draws are read as a K x N matrix by both sides (include/Utility.h:31-35)
  mmhost::ReferenceImplementation<Data_t, OperatorMap, OperatorReduce>(a.data(), b.data(), expected.data(), sz.n, sz.k,
                                                                       sz.m, kTransposedA);

  std::cout << "Running simulation...\n" << std::flush;
#ifdef MM_DYNAMIC_SIZES
  MatrixMultiplicationKernel(a.data(), b.data(), computed.data(), sz.n, sz.k, sz.m);
#else
  MatrixMultiplicationKernel(a.data(), b.data(), computed.data());
#endif
  std::cout << "Verifying results...\n" << std::flush;
  if (!mmhost::Verify(computed.data(), expected.data(), sz.n, sz.m)) return 1;
  std::cout << "Matrix-matrix multiplication successfully verified.\n";
  return 0;
}
