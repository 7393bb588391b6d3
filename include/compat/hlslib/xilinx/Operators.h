/*
 * hlslib/xilinx/Operators.h (include/compat) -- hlslib::op::{Add, Multiply, And, Min, Max}<T>: the (map, reduce) functors a
 * gemm_hls build names through MM_MAP_OP / MM_REDUCE_OP (include/Config.h.in:34-35) and its host applies in Naive
 * (include/Utility.h:29,37).  Each one also carries `code`, its mm_op_t in include/mm_gemm.h, which is how the
 * hlslib::ocl adapter (OpenCL.h next to this file) tells the device library which semiring the build was configured for.
 * Identities as documented in include/mm_gemm.h (Max: lowest()).
 */
#pragma once
#include <algorithm>
#include <limits>

#ifndef MM_GEMM_NO_KERNEL_SYMBOL
#define MM_GEMM_NO_KERNEL_SYMBOL /* a gemm_hls build declares MatrixMultiplicationKernel itself, with its pack types
                                    (include/MatrixMultiplication.h:155-171): that declaration is the one in force */
#endif
#include "mm_gemm.h"

namespace hlslib {
namespace op {

template <typename T>
struct Add {
  static constexpr mm_op_t code = MM_OP_ADD;
  static T Apply(T const &a, T const &b) { return a + b; }
  static constexpr T identity() { return T(0); }
};

template <typename T>
struct Multiply {
  static constexpr mm_op_t code = MM_OP_MULTIPLY;
  static T Apply(T const &a, T const &b) { return a * b; }
  static constexpr T identity() { return T(1); }
};

template <typename T>
struct And {
  static constexpr mm_op_t code = MM_OP_AND;
  static T Apply(T const &a, T const &b) { return T(a != T(0) && b != T(0)); }
  static constexpr T identity() { return T(1); }
};

template <typename T>
struct Min {
  static constexpr mm_op_t code = MM_OP_MIN;
  static T Apply(T const &a, T const &b) { return std::min(a, b); }
  static T identity() { return std::numeric_limits<T>::max(); }
};

template <typename T>
struct Max {
  static constexpr mm_op_t code = MM_OP_MAX;
  static T Apply(T const &a, T const &b) { return std::max(a, b); }
  static T identity() { return std::numeric_limits<T>::lowest(); }
};

}  // namespace op
}  // namespace hlslib
