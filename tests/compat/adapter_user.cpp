// A small hlslib::ocl client written for tests/test_compat_headers.py: the adapter's whole surface, including the forms
// the reference's host does not use (MemoryBank, the iterator-pair MakeBuffer, move of buffers, a wrong kernel name).
// Build: clang++ -std=c++17 -DMM_DYNAMIC_SIZES -Itests/compat -Iinclude/compat -Iinclude adapter_user.cpp -lmm_gemm_amd
#include <iostream>
#include <vector>

#include "hlslib/xilinx/DataPack.h"
#include "hlslib/xilinx/OpenCL.h"
#include "hlslib/xilinx/Utility.h"

int main(int argc, char **argv) {
  using namespace hlslib::ocl;
  const unsigned n = 70, k = 48, m = 64;
  using Pack = hlslib::DataPack<Data_t, 16>;
  static_assert(sizeof(Pack) == 16 * sizeof(Data_t), "a pack is its elements and nothing else");
  std::vector<Data_t> a(n * k), b(k * m), c(n * m, -1);
  for (unsigned i = 0; i < n * k; ++i) a[i] = (int)(i % 7) - 3;
  for (unsigned i = 0; i < k * m; ++i) b[i] = (int)(i % 5) - 2;
  std::vector<Pack, AlignedAllocator<Pack, 4096>> aPacked(n * k / 16), bPacked(k * m / 16), cPacked(n * m / 16);
  for (size_t i = 0; i < aPacked.size(); ++i) aPacked[i].Pack(&a[i * 16]);
  for (size_t i = 0; i < bPacked.size(); ++i) bPacked[i].Pack(&b[i * 16]);
  if (reinterpret_cast<uintptr_t>(aPacked.data()) % 4096 != 0) { std::cerr << "allocator alignment\n"; return 2; }
  try {
    Context context;
    auto program = context.MakeProgram("whatever.xclbin");
    auto aDevice = context.MakeBuffer<Pack, Access::read>(StorageType::DDR, 0, aPacked.cbegin(), aPacked.cend());
    auto bDevice = context.MakeBuffer<Pack, Access::read>(MemoryBank::bank1, bPacked.size());
    bDevice.CopyFromHost(bPacked.cbegin());
    auto cFirst = context.MakeBuffer<Pack, Access::write>(cPacked.size());
    Buffer<Pack, Access::write> cDevice(std::move(cFirst));
    bool refused = false;
    try { (void)program.MakeKernel("SomeOtherKernel", aDevice, bDevice, cDevice, n, k, m); } catch (ConfigurationError const &) { refused = true; }
    if (!refused) { std::cerr << "a wrong kernel name was accepted\n"; return 2; }
    if (argc > 1 && std::string(argv[1]) == "hw_emu") hlslib::SetEnvironmentVariable("XCL_EMULATION_MODE", "hw_emu");
    auto kernel = program.MakeKernel("MatrixMultiplicationKernel", aDevice, bDevice, cDevice, n, k, m);
    const auto elapsed = kernel.ExecuteTask();
    cDevice.CopyToHost(cPacked.begin());
    for (size_t i = 0; i < cPacked.size(); ++i) cPacked[i].Unpack(&c[i * 16]);
    for (unsigned i = 0; i < n; ++i)
      for (unsigned j = 0; j < m; ++j) {
        int want = 0;
        for (unsigned kk = 0; kk < k; ++kk) want += a[i * k + kk] * b[kk * m + j];
        if (c[i * m + j] != want) { std::cerr << "Mismatch at (" << i << ", " << j << ")\n"; return 1; }
      }
    std::cout << "adapter ok: " << kernel.Name() << " in " << elapsed.first << " s" << std::endl;
  } catch (std::runtime_error const &err) {
    std::cerr << "Execution failed with error: \"" << err.what() << "\"." << std::endl;
    return 1;
  }
  return 0;
}
