// TEST-ONLY shim for the subset of Xilinx ap_uint<W> used by kernel/Compute.cpp:161-165:
// W-bit unsigned counter with wrap-around, ++, assignment and integer conversion.
#pragma once
#include <cstdint>
#include "hls_half.h"  // Xilinx ap_int.h makes `half` visible too
template <int W>
class ap_uint {
  static_assert(W >= 0 && W <= 64, "shim supports up to 64 bits");
  static constexpr std::uint64_t Mask() {
    return W >= 64 ? ~std::uint64_t(0) : (W <= 0 ? std::uint64_t(1) : ((std::uint64_t(1) << W) - 1));
  }
  std::uint64_t v_ = 0;

 public:
  ap_uint() = default;
  ap_uint(std::uint64_t v) : v_(v & Mask()) {}
  ap_uint &operator=(std::uint64_t v) { v_ = v & Mask(); return *this; }
  operator std::uint64_t() const { return v_; }
  ap_uint &operator++() { v_ = (v_ + 1) & Mask(); return *this; }
  ap_uint operator++(int) { ap_uint t(*this); ++*this; return t; }
  ap_uint &operator--() { v_ = (v_ - 1) & Mask(); return *this; }
};
