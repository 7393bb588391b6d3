// Build-time configuration of the host binaries, the counterpart of the reference's generated
// Config.h (include/Config.h.in:15-35): -DMM_DATA_TYPE=<type> -DMM_MAP_OP=<Op> -DMM_REDUCE_OP=<Op>
// [-DMM_MEMORY_BUS_WIDTH_K=64 -DMM_MEMORY_BUS_WIDTH_M=64] [-DMM_DYNAMIC_SIZES | -DMM_SIZE_N=..]
// [-DMM_TRANSPOSED_A] [-DMM_POWER_METER].
// The device library itself is runtime-dispatched; these macros pick the configuration this
// binary drives, exactly one per binary as in the reference.
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>
#include <type_traits>

#include "mm_gemm.h"

#ifndef MM_DATA_TYPE
#define MM_DATA_TYPE float
#endif
#ifndef MM_MAP_OP
#define MM_MAP_OP Multiply
#endif
#ifndef MM_REDUCE_OP
#define MM_REDUCE_OP Add
#endif
#ifndef MM_MEMORY_BUS_WIDTH_K
#define MM_MEMORY_BUS_WIDTH_K 64
#endif
#ifndef MM_MEMORY_BUS_WIDTH_M
#define MM_MEMORY_BUS_WIDTH_M 64
#endif
#ifndef MM_MEMORY_BUS_WIDTH_N
#define MM_MEMORY_BUS_WIDTH_N 64
#endif

using half = _Float16;  // MM_DATA_TYPE=half (CMakeLists.txt:44-45)
using Data_t = MM_DATA_TYPE;

constexpr int kSeed = 5;  // include/MatrixMultiplication.h:14
constexpr int kMemoryWidthK = MM_MEMORY_BUS_WIDTH_K / sizeof(Data_t);  // MatrixMultiplication.h:18
constexpr int kMemoryWidthM = MM_MEMORY_BUS_WIDTH_M / sizeof(Data_t);  // MatrixMultiplication.h:24
// MM_TRANSPOSED_A (CMakeLists.txt:30,100-103): A is handed over as K x N; the host oracle indexes it
// that way (include/Utility.h:31-35) and the kernel's first parameter becomes MemoryPackN_t
// (include/MatrixMultiplication.h:46-66,156-162).
#ifdef MM_TRANSPOSED_A
constexpr bool kTransposedA = true;
constexpr int kMemoryWidthN = MM_MEMORY_BUS_WIDTH_N / sizeof(Data_t);  // MatrixMultiplication.h:49
#else
constexpr bool kTransposedA = false;
#endif
#ifndef MM_DYNAMIC_SIZES
#ifndef MM_SIZE_N
#define MM_SIZE_N 512
#endif
#ifndef MM_SIZE_K
#define MM_SIZE_K 512
#endif
#ifndef MM_SIZE_M
#define MM_SIZE_M 512
#endif
constexpr unsigned long kSizeN = MM_SIZE_N, kSizeK = MM_SIZE_K, kSizeM = MM_SIZE_M;
static_assert(kSizeK % kMemoryWidthK == 0, "K must be divisable by memory width.");
#endif

namespace mmhost {

// -DMM_MEMORY_TILE_SIZE_N=<BM> -DMM_MEMORY_TILE_SIZE_M=<BN> (the reference's build knob for the resident output tile,
// CMakeLists.txt:18-20 -> include/Config.h.in:19-23): pins this binary's kernel to a BM x BN tile WHEN the library has
// that geometry for the configuration -- float 128x256 / 256x256 / 128x128, double 256x128 / 128x128, half 256x256 /
// 128x256.  Any other value (e.g. the reference's 512x512) leaves the per-problem choice to the library, which is
// also what a build without the two macros does.  Returns the name of the knob it set, or nullptr.
inline const char *ApplyBuildTimeTile() {
#if defined(MM_MEMORY_TILE_SIZE_N) && defined(MM_MEMORY_TILE_SIZE_M)
  constexpr int bm = MM_MEMORY_TILE_SIZE_N, bn = MM_MEMORY_TILE_SIZE_M;
  struct Pin { int bm, bn; const char *knob; int value; };
  constexpr bool f32 = std::is_same<MM_DATA_TYPE, float>::value, f64 = std::is_same<MM_DATA_TYPE, double>::value,
                 f16 = std::is_same<MM_DATA_TYPE, _Float16>::value;
  const Pin pins[] = {{128, 256, f32 ? "f32_variant" : nullptr, 33}, {256, 256, f32 ? "f32_variant" : nullptr, 8},
                      {128, 128, f32 ? "f32_variant" : nullptr, 35}, {256, 128, f64 ? "f64_variant" : nullptr, 0},
                      {128, 128, f64 ? "f64_variant" : nullptr, 1},  {256, 256, f16 ? "f16_variant" : nullptr, -1},
                      {128, 256, f16 ? "f16_variant" : nullptr, 4}};
  for (const Pin &p : pins)
    if (p.knob && p.bm == bm && p.bn == bn) {
      if (p.value >= 0) mm_tuning_set(p.knob, p.value);
      return p.knob;
    }
#endif
  return nullptr;
}

template <typename T> struct DTypeOf;
#define MM_DTYPE_OF(T, E) template <> struct DTypeOf<T> { static constexpr mm_dtype_t value = E; };
MM_DTYPE_OF(float, MM_DTYPE_F32)
MM_DTYPE_OF(double, MM_DTYPE_F64)
MM_DTYPE_OF(half, MM_DTYPE_F16)
MM_DTYPE_OF(int8_t, MM_DTYPE_I8)
MM_DTYPE_OF(uint8_t, MM_DTYPE_U8)
MM_DTYPE_OF(int16_t, MM_DTYPE_I16)
MM_DTYPE_OF(uint16_t, MM_DTYPE_U16)
MM_DTYPE_OF(int32_t, MM_DTYPE_I32)
MM_DTYPE_OF(uint32_t, MM_DTYPE_U32)
MM_DTYPE_OF(long, MM_DTYPE_I64)
MM_DTYPE_OF(unsigned long, MM_DTYPE_U64)
#undef MM_DTYPE_OF

// half is neither integral nor floating_point for the standard traits, like the reference's half
template <typename T> struct IsHalf : std::is_same<T, half> {};

template <typename T> T HighestValue() {
  if constexpr (IsHalf<T>::value) return (T)65504.0f; else return std::numeric_limits<T>::max();
}
template <typename T> T LowestValue() {
  if constexpr (IsHalf<T>::value) return (T)-65504.0f; else return std::numeric_limits<T>::lowest();
}

// The (map, reduce) vocabulary of hlslib::op (absent third-party header; semantics as used at
// kernel/Compute.cpp:129,133 and include/Utility.h:29,37).
namespace op {
template <typename T> struct Add {
  static constexpr mm_op_t code = MM_OP_ADD;
  static T Apply(T a, T b) { return (T)(a + b); }
  static T identity() { return (T)0; }
};
template <typename T> struct Multiply {
  static constexpr mm_op_t code = MM_OP_MULTIPLY;
  static T Apply(T a, T b) { return (T)(a * b); }
  static T identity() { return (T)1; }
};
template <typename T> struct And {
  static constexpr mm_op_t code = MM_OP_AND;
  static T Apply(T a, T b) { return (T)((a != (T)0) && (b != (T)0)); }
  static T identity() { return (T)1; }
};
template <typename T> struct Min {
  static constexpr mm_op_t code = MM_OP_MIN;
  static T Apply(T a, T b) { return b < a ? b : a; }
  static T identity() { return HighestValue<T>(); }
};
template <typename T> struct Max {
  static constexpr mm_op_t code = MM_OP_MAX;
  static T Apply(T a, T b) { return a < b ? b : a; }
  static T identity() { return LowestValue<T>(); }
};
}  // namespace op

}  // namespace mmhost

using OperatorMap = mmhost::op::MM_MAP_OP<Data_t>;
using OperatorReduce = mmhost::op::MM_REDUCE_OP<Data_t>;
