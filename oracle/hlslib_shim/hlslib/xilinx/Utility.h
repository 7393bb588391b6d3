// TEST-ONLY shim: ConstLog2 and the environment helpers.
#pragma once
#include <cstdlib>
#include <string>
namespace hlslib {
// ceil(log2(x)) for x >= 1; ConstLog2(1) == 0.
constexpr unsigned ConstLog2(unsigned long x) { return x <= 1 ? 0 : 1 + ConstLog2((x + 1) / 2); }
inline void SetEnvironmentVariable(std::string const &k, std::string const &v) { setenv(k.c_str(), v.c_str(), 1); }
inline void UnsetEnvironmentVariable(std::string const &k) { unsetenv(k.c_str()); }
}  // namespace hlslib
