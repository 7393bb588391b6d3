// TEST-ONLY shim for the `half` type that Xilinx's ap_int.h/hls_half.h put into the global
// namespace (include/Utility.h:125-129 names it unconditionally; include/Config.h.in:8-10).
// IEEE binary16 storage, round-to-nearest-even conversions, arithmetic evaluated in double
// (exact for one +,-,* of two binary16 values) and rounded once.
#pragma once
#include <cmath>
#include <cstdint>
#include <iosfwd>
#include <limits>
class half {
  std::uint16_t bits_ = 0;
  static std::uint16_t FromDouble(double d) {
    std::uint16_t sign = 0;
    if (std::signbit(d)) { sign = 0x8000; d = -d; }
    if (std::isnan(d)) return sign | 0x7e00;
    if (d >= 65520.0) return sign | 0x7c00;
    if (d == 0.0) return sign;
    int e;
    (void)std::frexp(d, &e);
    int unb = e - 1;
    if (unb < -14) unb = -14;
    const double q = std::ldexp(d, 10 - unb);
    std::uint32_t m = static_cast<std::uint32_t>(std::nearbyint(q));
    int bexp = unb + 15;
    if (unb == -14 && m < 0x400) return sign | static_cast<std::uint16_t>(m);
    if (m == 0x800) { m = 0x400; bexp += 1; }
    if (bexp >= 31) return sign | 0x7c00;
    return sign | static_cast<std::uint16_t>((bexp << 10) | (m & 0x3ff));
  }
  double ToDouble() const {
    const int sign = bits_ >> 15, exp = (bits_ >> 10) & 0x1f, man = bits_ & 0x3ff;
    double v;
    if (exp == 0) v = std::ldexp(static_cast<double>(man), -24);
    else if (exp == 31) v = man ? std::numeric_limits<double>::quiet_NaN() : std::numeric_limits<double>::infinity();
    else v = std::ldexp(static_cast<double>(man | 0x400), exp - 25);
    return sign ? -v : v;
  }

 public:
  half() = default;
  half(double d) : bits_(FromDouble(d)) {}
  half(float f) : bits_(FromDouble(f)) {}
  half(int i) : bits_(FromDouble(i)) {}
  half(unsigned i) : bits_(FromDouble(i)) {}
  half(long i) : bits_(FromDouble(static_cast<double>(i))) {}
  half(unsigned long i) : bits_(FromDouble(static_cast<double>(i))) {}
  operator float() const { return static_cast<float>(ToDouble()); }
  std::uint16_t bits() const { return bits_; }
  friend half operator+(half a, half b) { return half(a.ToDouble() + b.ToDouble()); }
  friend half operator-(half a, half b) { return half(a.ToDouble() - b.ToDouble()); }
  friend half operator*(half a, half b) { return half(a.ToDouble() * b.ToDouble()); }
  friend half operator/(half a, half b) { return half(a.ToDouble() / b.ToDouble()); }
  half operator-() const { half h; h.bits_ = bits_ ^ 0x8000; return h; }
  half &operator+=(half o) { return *this = *this + o; }
  half &operator*=(half o) { return *this = *this * o; }
  friend bool operator==(half a, half b) { return a.ToDouble() == b.ToDouble(); }
  friend bool operator!=(half a, half b) { return !(a == b); }
  // exact matches for comparisons with integer literals (`diff != 0`, test/TestSimulation.cpp:84),
  // which are otherwise ambiguous between half(int) and operator float()
  friend bool operator==(half a, int b) { return a.ToDouble() == b; }
  friend bool operator!=(half a, int b) { return a.ToDouble() != b; }
  friend bool operator<(half a, half b) { return a.ToDouble() < b.ToDouble(); }
  friend bool operator>(half a, half b) { return b < a; }
  friend bool operator<=(half a, half b) { return !(b < a); }
  friend bool operator>=(half a, half b) { return !(a < b); }
};
namespace std {
inline half abs(half h) { return h < half(0.0) ? -h : h; }
template <> struct numeric_limits<half> {
  static constexpr bool is_specialized = true;
  static half max() { return half(65504.0); }
  static half lowest() { return half(-65504.0); }
  static half min() { return half(6.103515625e-05); }
};
}  // namespace std
