/*
 * hlslib/xilinx/DataPack.h (include/compat) -- hlslib::DataPack<T, W> as the reference's HOST code uses it:
 * include/MatrixMultiplication.h:21-32,52 (the MemoryPack*_t aliases), include/Utility.h:44-63 (Pack / Unpack:
 * result[i].Pack(&in[i * W]), in[i].Unpack(&out[i * W])).  W contiguous elements and nothing else, so an array of packs
 * IS the plain row-major Data_t array libmm_gemm_amd.so takes (include/mm_gemm.h).  hlslib itself is an un-vendored
 * submodule of the reference (.gitmodules:1-3); this is a stand-in written against those call sites.
 */
#pragma once
#include "hls_half.h" /* Xilinx's ap_int.h, which the original includes, makes the global `half` visible in every build */

namespace hlslib {

template <typename T, int width>
class DataPack {
  static_assert(width > 0, "DataPack width must be positive");

 public:
  static constexpr int kWidth = width;
  DataPack() : elements_{} {}
  explicit DataPack(T const &value) { Fill(value); }
  explicit DataPack(T const *source) { Pack(source); }
  T &operator[](int i) { return elements_[i]; }
  T const &operator[](int i) const { return elements_[i]; }
  T Get(int i) const { return elements_[i]; }
  void Set(int i, T const &value) { elements_[i] = value; }
  void Fill(T const &value) {
    for (T &e : elements_) e = value;
  }
  void Pack(T const *source) {
    for (int i = 0; i < width; ++i) elements_[i] = source[i];
  }
  void Unpack(T *destination) const {
    for (int i = 0; i < width; ++i) destination[i] = elements_[i];
  }

 private:
  T elements_[width];
};

}  // namespace hlslib
