// TestSimulation -- the role of the reference's CTest binary (test/TestSimulation.cpp) with the GPU
// behind the symbol it calls.  Command line, size checks and messages are the reference's:
//     TestSimulation N K M        -> "Running simulation...", "Verifying results...",
//                                    "Matrix-matrix multiplication successfully verified."
// The single call MatrixMultiplicationKernel(a, b, c, N, K, M) takes HOST pointers; in the
// reference it runs the HLS kernel as host threads, here libmm_gemm_amd.so exports the symbol and
// it runs on MI355X device 0 with this build's MM_DATA_TYPE / MM_MAP_OP / MM_REDUCE_OP.
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "HostReference.h"

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
  mmhost::ReferenceImplementation<Data_t, OperatorMap, OperatorReduce>(a.data(), b.data(), expected.data(), sz.n, sz.k,
                                                                       sz.m);

  const mm_config_t config = {mmhost::DTypeOf<Data_t>::value, OperatorMap::code, OperatorReduce::code, MM_PATH_AUTO,
                              MM_A_ROW_MAJOR};
  if (mm_set_default_config(&config) != MM_OK) {
    std::cerr << mm_last_error() << std::endl;
    return 1;
  }
  std::cout << "Running simulation...\n" << std::flush;
  MatrixMultiplicationKernel(a.data(), b.data(), computed.data(), sz.n, sz.k, sz.m);
  std::cout << "Verifying results...\n" << std::flush;
  if (!mmhost::Verify(computed.data(), expected.data(), sz.n, sz.m)) return 1;
  std::cout << "Matrix-matrix multiplication successfully verified.\n";
  return 0;
}
