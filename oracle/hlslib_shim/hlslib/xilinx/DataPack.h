// TEST-ONLY shim: fixed-width vector of T with the members the reference uses.
#pragma once
#include "ap_int.h"
namespace hlslib {
template <typename T, int width>
class DataPack {
  static_assert(width > 0, "width must be positive");
  T data_[width];

 public:
  static constexpr int kWidth = width;
  DataPack() : data_{} {}
  explicit DataPack(T const &fill) { for (int i = 0; i < width; ++i) data_[i] = fill; }
  explicit DataPack(T const *in) { Pack(in); }
  T &operator[](int i) { return data_[i]; }
  T const &operator[](int i) const { return data_[i]; }
  T Get(int i) const { return data_[i]; }
  void Set(int i, T const &v) { data_[i] = v; }
  void Fill(T const &v) { for (int i = 0; i < width; ++i) data_[i] = v; }
  void Pack(T const *in) { for (int i = 0; i < width; ++i) data_[i] = in[i]; }
  void Unpack(T *out) const { for (int i = 0; i < width; ++i) out[i] = data_[i]; }
  void operator<<(T const *in) { Pack(in); }
  void operator>>(T *out) const { Unpack(out); }
};
}  // namespace hlslib
