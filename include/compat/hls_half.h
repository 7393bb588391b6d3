/*
 * hls_half.h -- the global `half` type a gemm_hls build sees (Vitis puts it on the include path; the reference names it
 * at include/Config.h.in:8-10 under MM_HALF_PRECISION and, unconditionally, at include/Utility.h:125-129).
 *
 * Part of include/compat: the headers that let the reference's OWN host sources (host/RunHardware.cpp,
 * test/TestSimulation.cpp, src/PrintSpecifications.cpp) compile unmodified against libmm_gemm_amd.so.
 * IEEE binary16 on the compiler's native _Float16 (clang / amdclang++): every +, -, *, / rounds to binary16 once, which is
 * what the reference's Naive relies on when Data_t = half.  A class, not a typedef, on purpose: like Xilinx's type it
 * is NOT std::is_floating_point, so the reference's verification loops compare half results EXACTLY
 * (host/RunHardware.cpp:212-217, test/TestSimulation.cpp:80-85).
 */
#pragma once
#include <cmath>
#include <limits>
#include <ostream>

class half {
 public:
  half() = default;
  half(double v) : v_((_Float16)v) {}
  half(float v) : v_((_Float16)v) {}
  half(int v) : v_((_Float16)v) {}
  half(unsigned v) : v_((_Float16)v) {}
  half(long v) : v_((_Float16)v) {}
  half(unsigned long v) : v_((_Float16)v) {}
  operator float() const { return (float)v_; }
  _Float16 native() const { return v_; }

  friend half operator+(half a, half b) { return from(a.v_ + b.v_); }
  friend half operator-(half a, half b) { return from(a.v_ - b.v_); }
  friend half operator*(half a, half b) { return from(a.v_ * b.v_); }
  friend half operator/(half a, half b) { return from(a.v_ / b.v_); }
  half operator-() const { return from(-v_); }
  half &operator+=(half o) { v_ = v_ + o.v_; return *this; }
  half &operator-=(half o) { v_ = v_ - o.v_; return *this; }
  half &operator*=(half o) { v_ = v_ * o.v_; return *this; }
  friend bool operator==(half a, half b) { return a.v_ == b.v_; }
  friend bool operator!=(half a, half b) { return a.v_ != b.v_; }
  friend bool operator<(half a, half b) { return a.v_ < b.v_; }
  friend bool operator>(half a, half b) { return a.v_ > b.v_; }
  friend bool operator<=(half a, half b) { return a.v_ <= b.v_; }
  friend bool operator>=(half a, half b) { return a.v_ >= b.v_; }
  /* `diff != 0` in the reference's verification loops: an exact overload, or half(int) and operator float() tie */
  friend bool operator==(half a, int b) { return (float)a.v_ == (float)b; }
  friend bool operator!=(half a, int b) { return (float)a.v_ != (float)b; }
  friend std::ostream &operator<<(std::ostream &os, half h) { return os << (float)h.v_; }

 private:
  static half from(_Float16 v) { half h; h.v_ = v; return h; }
  _Float16 v_ = (_Float16)0;
};
static_assert(sizeof(half) == 2, "half must be layout-compatible with binary16 storage");

namespace std {
inline half abs(half h) { return h < half(0) ? -h : h; }
template <> struct numeric_limits<half> {
  static constexpr bool is_specialized = true;
  static half max() { return half(65504.0); }
  static half lowest() { return half(-65504.0); }
  static half min() { return half(6.103515625e-05); }
};
}  // namespace std
