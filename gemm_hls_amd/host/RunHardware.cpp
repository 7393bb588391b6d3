// RunHardware -- host runner with the command line, checks, messages, GOp/s report and exit codes
// of the reference's host/RunHardware.cpp, driving the MI355X library through its C ABI
// (include/mm_gemm.h) instead of hlslib's OpenCL wrapper.
//
//   RunHardware.exe N K M [<mode [hw/hw_emu]>] [<verify [on/off]>]      (MM_DYNAMIC_SIZES)
//   RunHardware.exe [<mode [hw/hw_emu]>] [<verify [on/off]>]            (static sizes)
//
//   hw      the fast kernel family for this build's (MM_DATA_TYPE, MM_MAP_OP, MM_REDUCE_OP)
//   hw_emu  the k-ordered kernel that is bit-identical to the reference's Naive (MM_PATH_ORDERED);
//           like the reference's hw_emu it executes the same contract more slowly, on the device
//   on/off  verify against ReferenceImplementation (BLAS / Naive) or time only.  With `off` the
//           reference leaves device memory uninitialised; here it is filled on the device with
//           the same value distribution so that clocks behave like a real run.
// Build flags (the reference's CMake options of the same names, CMakeLists.txt:21-36): MM_DYNAMIC_SIZES
// / MM_SIZE_{N,K,M}; MM_TRANSPOSED_A (A generated, handed over and verified as K x N); MM_POWER_METER
// (sample the GPU's power while the kernel runs and print the reference's "Measured an average power
// of ... W" line, host/RunHardware.cpp:156-172,182-185 -- the board's own sensor through
// librocm_smi64 / hwmon instead of the reference's PSU meter).
// Environment: MM_GPUS=<g> splits the rows of C over g devices (no collective; default 1).
//              MM_PATH=split runs "hw" through MM_PATH_SPLIT (float (Multiply, Add) builds only: fp32 on the
//              bf16 matrix cores from three planes per operand, include/mm_gemm.h); anything else is refused.
#include <chrono>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "HostReference.h"
#ifdef MM_POWER_METER
#include "PowerMeter.h"
#endif

namespace {

void PrintUsage() {
#ifndef MM_DYNAMIC_SIZES
  std::cerr << "Usage: ./RunHardware.exe <mode [hw/hw_emu]> [<verify [on/off]>]\n" << std::flush;
#else
  std::cerr << "Usage: ./RunHardware.exe N K M [<mode [hw/hw_emu]>] [<verify [on/off]>]\n" << std::flush;
#endif
}

void Check(int status, const char *what) {
  if (status != MM_OK) throw std::runtime_error(std::string(what) + ": " + mm_last_error());
}

// RAII device buffer (the role of hlslib::ocl::Buffer, host/RunHardware.cpp:122-138)
struct DeviceBuffer {
  int device;
  void *ptr = nullptr;
  size_t bytes;
  DeviceBuffer(int dev, size_t nbytes) : device(dev), bytes(nbytes) { Check(mm_alloc(dev, nbytes, &ptr), "mm_alloc"); }
  ~DeviceBuffer() { if (ptr) mm_free(device, ptr); }
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};

}  // namespace

int main(int argc, char **argv) {
  mmhost::ApplyBuildTimeTile();   // -DMM_MEMORY_TILE_SIZE_N/M: pin the resident tile if the library has that geometry
#ifdef MM_HALF_CONTRACT_REFERENCE
  // a half build that keeps the reference's arithmetic under "hw" too (binary16 accumulating in binary16,
  // kernel/Compute.cpp:129-133); the same as MM_HALF_CONTRACT=reference in the environment
  mm_tuning_set("half_contract", 1);
#endif
  bool emulation = false;
  bool verify = true;
#ifdef MM_DYNAMIC_SIZES
  if (argc > 6 || argc < 4) {
    PrintUsage();
    return 1;
  }
  unsigned size_n, size_k, size_m;
  try {
    size_n = std::stoul(argv[1]);
    size_k = std::stoul(argv[2]);
    size_m = std::stoul(argv[3]);
  } catch (std::exception const &) {
    PrintUsage();
    return 1;
  }
  int next_arg = 4;
  if (size_k % kMemoryWidthK != 0) {
    std::cerr << "K (" << size_k << ") must be divisable by the memory width in K (" << kMemoryWidthK << ")." << std::endl;
    return 1;
  }
  if (size_m % kMemoryWidthM != 0) {
    std::cerr << "M (" << size_m << ") must be divisable by the memory width in M (" << kMemoryWidthM << ")." << std::endl;
    return 1;
  }
#else
  if (argc > 3) {
    PrintUsage();
    return 1;
  }
  constexpr unsigned size_n = kSizeN, size_k = kSizeK, size_m = kSizeM;
  int next_arg = 1;
#endif
  if (next_arg < argc) {
    const std::string mode(argv[next_arg++]);
    if (mode == "hw_emu") {
      emulation = true;
    } else if (mode != "hw") {
      PrintUsage();
      return 1;
    }
  }
  if (next_arg < argc) {
    const std::string v(argv[next_arg++]);
    if (v == "off") {
      verify = false;
    } else if (v != "on") {
      PrintUsage();
      return 1;
    }
  }
  const int gpus = std::getenv("MM_GPUS") ? std::max(1, std::atoi(std::getenv("MM_GPUS"))) : 1;

  mm_path_t path = emulation ? MM_PATH_ORDERED : MM_PATH_AUTO;
  if (const char *want = std::getenv("MM_PATH")) {
    if (std::string(want) == "split" && !emulation) {
      path = MM_PATH_SPLIT;
    } else if (std::string(want) != "auto") {
      std::cerr << "MM_PATH must be \"auto\" or \"split\" (with hw); got \"" << want << "\"\n";
      return 1;
    }
  }
  const mm_config_t cfg = {mmhost::DTypeOf<Data_t>::value, OperatorMap::code, OperatorReduce::code, path,
                           kTransposedA ? MM_A_TRANSPOSED : MM_A_ROW_MAJOR};
  const size_t count_a = (size_t)size_n * size_k, count_b = (size_t)size_k * size_m, count_c = (size_t)size_n * size_m;

  std::vector<Data_t> a, b, cRef, cTest;
  std::cout << "Initializing host memory..." << std::flush;
  if (verify || gpus > 1) {
    // The reference's generator: ONE engine seeded with kSeed, real or integer uniform on [1, 10],
    // all of A drawn first, then all of B (host/RunHardware.cpp:31-35,99-105).  With MM_TRANSPOSED_A
    // the same N*K draws ARE the K x N matrix (the reference fills one flat vector and lets the
    // kernel and Naive index it as a[k * N + n], include/Utility.h:31-35).
    std::default_random_engine rng(kSeed);
    typename std::conditional<std::is_integral<Data_t>::value, std::uniform_int_distribution<unsigned long>,
                              std::uniform_real_distribution<double>>::type dist(1, 10);
    a.resize(count_a);
    b.resize(count_b);
    for (auto &x : a) x = Data_t(dist(rng));
    for (auto &x : b) x = Data_t(dist(rng));
    cTest.assign(count_c, Data_t(0));
    if (verify) cRef.assign(count_c, Data_t(0));
  }
  std::cout << " Done.\n";

  try {
    std::cout << "Initializing HIP context...\n" << std::flush;
    int device_count = 0;
    Check(mm_init(&device_count), "mm_init");
    {  // MM_GPUS beyond the visible devices is an error -- unless the library was told to deal logical devices out over
       // the physical ones (MM_MD_VIRTUAL_DEVICES, tests on a 1-GPU box), which mm_gemm_multi_device then bounds itself
      int virt = -1;
      (void)mm_tuning_get("md_virtual_devices", &virt);
      if (gpus > std::max(device_count, virt)) throw std::runtime_error("MM_GPUS exceeds the number of visible devices");
    }

    double elapsed = 0.0;
#ifdef MM_POWER_METER
    double average_power = 0.0;
    size_t power_samples = 0;
    std::string power_source = "no single-device launch", power_bdf = "?";
#endif
    if (gpus > 1) {
      std::cout << "Initializing device memory on " << gpus << " devices, copying row slabs of A and C and replicas of B...\n"
                << std::flush;
      std::cout << "Executing kernel...\n" << std::flush;
      std::vector<double> per_device(gpus, 0.0);
      double host_wall = 0.0;
      Check(mm_gemm_multi_device_timed(gpus, &cfg, a.data(), b.data(), cTest.data(), size_n, size_k, size_m, &elapsed,
                                       per_device.data(), &host_wall),
            "mm_gemm_multi_device_timed");
      // where the time went: each device's own kernel time (HIP events on its stream); the job's time is their maximum
      for (int g = 0; g < gpus; ++g) {
        unsigned row0 = 0, rows = 0;
        Check(mm_row_slab(&cfg, size_n, size_k, size_m, gpus, g, &row0, &rows), "mm_row_slab");
        std::cout << "  device " << g << ": rows [" << row0 << ", " << row0 + rows << ") in " << per_device[g] << " seconds"
                  << (per_device[g] == elapsed && rows ? "  <- slowest" : "") << "\n";
      }
      std::cout << "  host clock, first dispatch to last completion: " << host_wall << " seconds\n" << std::flush;
    } else {
      std::cout << "Initializing device memory...\n" << std::flush;
      DeviceBuffer aDevice(0, count_a * sizeof(Data_t)), bDevice(0, count_b * sizeof(Data_t)),
          cDevice(0, count_c * sizeof(Data_t));
      if (verify) {
        std::cout << "Copying memory to device...\n" << std::flush;
        Check(mm_copy_to_device(0, aDevice.ptr, a.data(), aDevice.bytes), "copy A");
        Check(mm_copy_to_device(0, bDevice.ptr, b.data(), bDevice.bytes), "copy B");
        Check(mm_copy_to_device(0, cDevice.ptr, cTest.data(), cDevice.bytes), "copy C");
      } else {
        Check(mm_fill_device(0, cfg.dtype, aDevice.ptr, count_a, 1), "fill A");
        Check(mm_fill_device(0, cfg.dtype, bDevice.ptr, count_b, 2), "fill B");
      }
      std::cout << "Creating kernel...\n" << std::flush;
      std::cout << "Executing kernel (" << mm_kernel_name(&cfg, size_n, size_k, size_m) << ")...\n" << std::flush;
      if (!verify) {  // timing run: one untimed launch first so that code upload is not in the number
        Check(mm_gemm_launch(0, &cfg, aDevice.ptr, bDevice.ptr, cDevice.ptr, size_n, size_k, size_m, nullptr), "warm-up");
      }
#ifdef MM_POWER_METER
      char bdf[32] = {0};
      if (mm_device_pci_bus_id(0, bdf, sizeof bdf) != MM_OK) bdf[0] = 0;
      mmhost::PowerMeter pm(0, 2, bdf);  // HIP device 0 by PCI address, 2 ms sampling period (the reference samples its PSU every 10 ms)
      pm.Start();
#endif
      Check(mm_gemm_launch(0, &cfg, aDevice.ptr, bDevice.ptr, cDevice.ptr, size_n, size_k, size_m, &elapsed),
            "mm_gemm_launch");
#ifdef MM_POWER_METER
      // A GPU kernel is over in milliseconds, shorter than the sensor's own averaging window: keep the
      // same launch running (untimed) until at least MM_POWER_WINDOW_MS (default 500) were sampled.
      {
        const char *w = std::getenv("MM_POWER_WINDOW_MS");
        const double window = (w ? std::atof(w) : 500.0) * 1e-3;
        while (pm.Elapsed() < window)
          Check(mm_gemm_launch(0, &cfg, aDevice.ptr, bDevice.ptr, cDevice.ptr, size_n, size_k, size_m, nullptr), "power window");
      }
      pm.Stop();
      average_power = pm.Average();
      power_samples = pm.Samples();
      power_source = pm.Source();
      power_bdf = bdf[0] ? bdf : "unknown";
#endif
      if (verify) {
        std::cout << "Copying back result...\n" << std::flush;
        Check(mm_copy_to_host(0, cTest.data(), cDevice.ptr, cDevice.bytes), "copy back");
      }
    }
    const auto perf = 1e-9 * (2 * static_cast<float>(size_n) * size_k * size_m) / elapsed;
    // Same line as the reference (host/RunHardware.cpp:178-180).  Default stream formatting turns
    // 1.2e6 GOp/s into "1.2e+06", which the reference's own parser (scripts/build_manager.py:601,
    // "([\\d\\.]+) seconds[^\\d]+([\\d\\.]+) GOp/s") cannot read, so large values print in fixed notation.
    std::cout << "Kernel executed in " << elapsed << " seconds, corresponding to a performance of ";
    if (perf >= 1e6) {
      const auto flags = std::cout.flags();
      const auto prec = std::cout.precision();
      std::cout << std::fixed << std::setprecision(0) << perf;
      std::cout.flags(flags);
      std::cout.precision(prec);
    } else {
      std::cout << perf;
    }
    std::cout << " GOp/s.\n";
#ifdef MM_POWER_METER
    // the reference's line (host/RunHardware.cpp:182-185) up to "W": scripts/build_manager.py:603-604
    // parses "([\\d\\.]+) W"; what was measured is this GPU's board power, not "the full system"
    std::cout << "Measured an average power of " << average_power << " W for the GPU (" << power_samples << " samples, "
              << power_source << ", PCI " << power_bdf << ").\n";
#endif
  } catch (std::runtime_error const &err) {
    std::cerr << "Execution failed with error: \"" << err.what() << "\"." << std::endl;
    return 1;
  }

  if (verify) {
    std::cout << "Running reference implementation...\n" << std::flush;
    const auto t0 = std::chrono::steady_clock::now();
    // hw_emu runs the k-ordered kernel, which follows the reference's Naive to the bit -- for half
    // that means accumulating in binary16, so it is checked against exactly that, exactly
    // -- and so does "hw" when the process asked for the reference's half contract (MM_HALF_CONTRACT=reference /
    // -DMM_HALF_CONTRACT_REFERENCE): timed AND verified on the same kernel, with the reference's exact comparison
    int half_contract = -1;
    (void)mm_tuning_get("half_contract", &half_contract);
    const bool reference_half = std::is_same<OperatorMap, mmhost::op::Multiply<Data_t>>::value &&
                                std::is_same<OperatorReduce, mmhost::op::Add<Data_t>>::value && half_contract == 1;
    const bool half_emulation = (emulation || reference_half) && mmhost::IsHalf<Data_t>::value;
    mmhost::ReferenceImplementation<Data_t, OperatorMap, OperatorReduce>(a.data(), b.data(), cRef.data(), size_n, size_k,
                                                                         size_m, kTransposedA, half_emulation);
    const double tref = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Reference implementation took " << tref << " seconds on " << mmhost::HostThreads() << " host threads.\n";
    std::cout << "Verifying result...\n" << std::flush;
    if (!mmhost::Verify(cTest.data(), cRef.data(), size_n, size_m, half_emulation)) return 1;
    std::cout << "Successfully verified." << std::endl;
  }
  return 0;
}
