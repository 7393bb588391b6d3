// TEST-ONLY shim of hlslib::op (absent third-party header).  Semantics restated from the
// published library and pinned by the reference's own use: (Multiply, Add) must equal
// cblas_sgemm/dgemm (include/Utility.h:76-103); identity() seeds Naive (include/Utility.h:29).
#pragma once
#include <algorithm>
#include <limits>
namespace hlslib {
namespace op {
template <typename T> struct Add {
  static T Apply(T const &a, T const &b) { return a + b; }
  static constexpr T identity() { return T(0); }
};
template <typename T> struct Multiply {
  static T Apply(T const &a, T const &b) { return a * b; }
  static constexpr T identity() { return T(1); }
};
template <typename T> struct And {
  static T Apply(T const &a, T const &b) { return a && b; }
  static constexpr T identity() { return T(1); }
};
template <typename T> struct Min {
  static T Apply(T const &a, T const &b) { return std::min(a, b); }
  static constexpr T identity() { return std::numeric_limits<T>::max(); }
};
template <typename T> struct Max {
  static T Apply(T const &a, T const &b) { return std::max(a, b); }
  static constexpr T identity() { return std::numeric_limits<T>::lowest(); }
};
}  // namespace op
}  // namespace hlslib
