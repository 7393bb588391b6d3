/*
 * hlslib/xilinx/Utility.h (include/compat) -- the helpers the reference's host calls: the environment setters through
 * which host/RunHardware.cpp:39,76 selects hardware emulation (XCL_EMULATION_MODE=hw_emu, which the hlslib::ocl adapter
 * next to this file reads back as "run the k-ordered kernel"), and the integer helpers its headers use.
 */
#pragma once
#include <cstdlib>
#include <string>

namespace hlslib {

inline void SetEnvironmentVariable(std::string const &name, std::string const &value) {
  ::setenv(name.c_str(), value.c_str(), 1);
}
inline void UnsetEnvironmentVariable(std::string const &name) { ::unsetenv(name.c_str()); }

template <typename T>
constexpr T CeilDivide(T a, T b) {
  return (a + b - 1) / b;
}
/* smallest e with 2^e >= x */
constexpr unsigned ConstLog2(unsigned long x) { return x <= 1 ? 0u : 1u + ConstLog2((x + 1) / 2); }

}  // namespace hlslib
