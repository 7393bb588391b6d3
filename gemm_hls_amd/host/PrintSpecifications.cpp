// PrintSpecifications -- the MI355X counterpart of the reference's src/PrintSpecifications.cpp:
// prints, for this build's (MM_DATA_TYPE, MM_MAP_OP, MM_REDUCE_OP) and a problem size, the
// operation count, the expected and ideal runtime/performance at a given clock, the tile
// geometry and the communication-volume model.  Same command line:
//     PrintSpecifications N K M [<frequency MHz>]            (MM_DYNAMIC_SIZES)
// "Ideal" = every compute unit issuing its matrix/vector instruction every cycle
// (2 * ops/clk/CU * CUs * f, the GPU reading of 2 * P_N * P_M * f, README.md:62-66).
// "Expected" adds what the kernel's structure costs on top: whole output tiles (ragged edges are
// computed in full), whole waves of workgroups over the 256 CUs, and the per-tile prologue
// (first k-slab fetch) and epilogue (C tile write) during which the matrix cores idle -- the role
// the drain term iN*(P_M*iM + P_N*iM) plays in the reference's model (:45-50).
// It needs no GPU: geometry comes from mm_kernel_info().
#include <cmath>
#include <iostream>
#include <string>

#include "HostConfig.h"

namespace {

// What the command line asks for.  The reference's grammar (src/PrintSpecifications.cpp:16-36): three sizes first when
// the build has dynamic sizes, then an optional frequency in MHz; anything else prints the usage line and exits 1.
struct Request {
  unsigned n = 0, k = 0, m = 0;
  bool has_frequency = false;
  float frequency_mhz = 0;
};

#ifdef MM_DYNAMIC_SIZES
constexpr int kSizeArgs = 3;
constexpr const char *kUsageTail = " N K M [<routed_frequency>]\n";
#else
constexpr int kSizeArgs = 0;
constexpr const char *kUsageTail = " [<routed frequency>]\n";
#endif

bool Parse(int argc, char **argv, Request *req) {
  const int given = argc - 1;
  if (given < kSizeArgs || given > kSizeArgs + 1) return false;
#ifdef MM_DYNAMIC_SIZES
  req->n = std::stoul(argv[1]);
  req->k = std::stoul(argv[2]);
  req->m = std::stoul(argv[3]);
#else
  req->n = kSizeN;
  req->k = kSizeK;
  req->m = kSizeM;
#endif
  if (given == kSizeArgs + 1) {
    req->has_frequency = true;
    req->frequency_mhz = std::stof(argv[kSizeArgs + 1]);
  }
  return true;
}

}  // namespace

int main(int argc, char **argv) {
  Request req;
  if (!Parse(argc, argv, &req)) {
    std::cerr << "Usage: " << argv[0] << kUsageTail << std::flush;
    return 1;
  }
  mmhost::ApplyBuildTimeTile();
  const unsigned size_n = req.n, size_k = req.k, size_m = req.m;
  const mm_config_t cfg = {mmhost::DTypeOf<Data_t>::value, OperatorMap::code, OperatorReduce::code, MM_PATH_AUTO,
                           MM_A_ROW_MAJOR};
  mm_kernel_info_t info;
  if (mm_kernel_info(&cfg, size_n, size_k, size_m, &info) != MM_OK) {
    std::cerr << mm_last_error() << "\n";
    return 1;
  }
  const float frequency = req.has_frequency ? req.frequency_mhz : (float)info.max_clock_mhz;

  const unsigned long long nOps = 2ull * size_n * size_k * size_m;
  const unsigned long long tilesN = (size_n + info.tile_n - 1) / info.tile_n, tilesM = (size_m + info.tile_m - 1) / info.tile_m;
  const double hz = 1e6 * frequency;
  const double ideal_perf = 1e-9 * info.ops_per_clk_per_cu * info.compute_units * hz;  // GOp/s
  const double ideal_runtime = nOps / (1e9 * ideal_perf);
  // per workgroup: K-loop cycles at full issue rate + prologue/epilogue with idle matrix cores
  const double tile_ops = 2.0 * info.tile_n * info.tile_m * size_k;
  const double eff = info.measured_issue_efficiency > 0 ? info.measured_issue_efficiency : 1.0;
  const double loop_cycles = tile_ops / (info.ops_per_clk_per_cu * eff);  // measured sustained issue rate
  const double hbm_bytes_per_clk_per_cu = 6.3e12 / (info.compute_units * hz);  // achievable HBM rate, shared
  const double prologue_cycles = 2000.0;  // ~ one HBM round trip + first fragment reads
  const double epilogue_cycles = (double)info.tile_n * info.tile_m * sizeof(Data_t) / hbm_bytes_per_clk_per_cu;
  const double waves_of_workgroups = std::ceil((double)(tilesN * tilesM) / info.compute_units);
  const double expected_runtime = waves_of_workgroups * (loop_cycles + prologue_cycles + epilogue_cycles) / hz;
  const double expected_perf = 1e-9 * nOps / expected_runtime;

  std::cout << "Kernel:               " << mm_kernel_name(&cfg, size_n, size_k, size_m) << "\n";
  std::cout << "Frequency:            " << frequency << " MHz\n";
  std::cout << "Number of operations: " << nOps << " (" << static_cast<float>(nOps) << ")\n";
  std::cout << "Expected runtime:     " << expected_runtime << " seconds\n";
  std::cout << "Ideal runtime:        " << ideal_runtime << " seconds\n";
  std::cout << "Percentage of deal:   " << 100 * ideal_runtime / expected_runtime << "%\n";
  std::cout << "Expected performance: " << expected_perf << " GOp/s\n";
  std::cout << "Ideal performance:    " << ideal_perf << " GOp/s\n";
  std::cout << "Compute tiles: " << info.inst_n << "x" << info.inst_m << "x" << info.inst_k << " per instruction, "
            << info.wavefronts << " wavefronts per workgroup, " << info.compute_units << " compute units ("
            << info.ops_per_clk_per_cu * info.compute_units / 2 << " parallel adders/multipliers)\n";
  std::cout << "Memory tile size: " << info.tile_n << "x" << info.tile_m << " (k-slab " << info.tile_k << ")\n";
  std::cout << "Tiles in N (outer/inner): " << tilesN << " / " << info.tile_n / (info.inst_n > 1 ? info.inst_n : 1) << "\n";
  std::cout << "Tiles in M (outer/inner): " << tilesM << " / " << info.tile_m / (info.inst_m > 1 && info.inst_n > 1 ? info.inst_m : 1)
            << "\n";
  const unsigned long long communicationVolume =
      (unsigned long long)size_n * size_m * (1 + size_k / info.tile_n + size_k / info.tile_m);
  std::cout << "Communication volume: " << communicationVolume << "\n";
  const double ioAccesses = communicationVolume / (3 * static_cast<double>(size_n) * size_m * size_k);
  std::cout << "I/O access fraction: " << ioAccesses << "\n";
  return 0;
}
