// TestSimulation -- the reference's CTest binary (test/TestSimulation.cpp) against the GPU:
// same command line (`TestSimulation N K M`), same size checks and messages, seeded inputs,
// ONE direct call of the reference's entry point
//     MatrixMultiplicationKernel(a, b, c, size_n, size_k, size_m)          (host pointers)
// and the same comparison against ReferenceImplementation.  In the reference that call runs the
// HLS kernel as host threads; here the symbol is exported by libmm_gemm_amd.so and runs on the
// MI355X (device 0), configured by this build's MM_DATA_TYPE / MM_MAP_OP / MM_REDUCE_OP.
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "HostReference.h"

int main(int argc, char **argv) {
#ifdef MM_DYNAMIC_SIZES
  if (argc < 4 || argc > 4) {
    std::cerr << "Usage: ./TestSimulation N K M" << std::endl;
    return 1;
  }
  const unsigned size_n = std::stoul(argv[1]), size_k = std::stoul(argv[2]), size_m = std::stoul(argv[3]);
  if (size_k % kMemoryWidthK != 0) {
    std::cerr << "K must be divisable by memory width." << std::endl;
    return 1;
  }
  if (size_m % kMemoryWidthM != 0) {
    std::cerr << "M must be divisable by memory width." << std::endl;
    return 1;
  }
#else
  constexpr unsigned size_n = kSizeN, size_k = kSizeK, size_m = kSizeM;
#endif
  std::vector<Data_t> a((size_t)size_n * size_k), b((size_t)size_k * size_m);
  std::vector<Data_t> cReference((size_t)size_n * size_m, Data_t(0)), cKernel((size_t)size_n * size_m, Data_t(0));
  std::default_random_engine rng(kSeed);
  typename std::conditional<std::is_integral<Data_t>::value, std::uniform_int_distribution<unsigned long>,
                            std::uniform_real_distribution<double>>::type dist(1, 10);
  for (auto &x : a) x = Data_t(dist(rng));
  for (auto &x : b) x = Data_t(dist(rng));

  mmhost::ReferenceImplementation<Data_t, OperatorMap, OperatorReduce>(a.data(), b.data(), cReference.data(), size_n,
                                                                       size_k, size_m);
  const mm_config_t cfg = {mmhost::DTypeOf<Data_t>::value, OperatorMap::code, OperatorReduce::code, MM_PATH_AUTO,
                           MM_A_ROW_MAJOR};
  if (mm_set_default_config(&cfg) != MM_OK) {
    std::cerr << mm_last_error() << std::endl;
    return 1;
  }
  std::cout << "Running simulation...\n" << std::flush;
  MatrixMultiplicationKernel(a.data(), b.data(), cKernel.data(), size_n, size_k, size_m);
  std::cout << "Verifying results...\n" << std::flush;
  if (!mmhost::Verify(cKernel.data(), cReference.data(), size_n, size_m)) return 1;
  std::cout << "Matrix-matrix multiplication successfully verified.\n";
  return 0;
}
