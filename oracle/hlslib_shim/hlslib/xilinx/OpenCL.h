// TEST-ONLY shim: only the aligned allocator that include/Utility.h:48 names.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <new>
namespace hlslib {
namespace ocl {
template <typename T, std::size_t alignment>
struct AlignedAllocator {
  using value_type = T;
  template <typename U> struct rebind { using other = AlignedAllocator<U, alignment>; };
  AlignedAllocator() = default;
  template <typename U> AlignedAllocator(AlignedAllocator<U, alignment> const &) {}
  T *allocate(std::size_t n) {
    void *p = nullptr;
    if (posix_memalign(&p, alignment, n * sizeof(T) ? n * sizeof(T) : alignment)) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, std::size_t) { free(p); }
  template <typename U> bool operator==(AlignedAllocator<U, alignment> const &) const { return true; }
  template <typename U> bool operator!=(AlignedAllocator<U, alignment> const &) const { return false; }
};
}  // namespace ocl
}  // namespace hlslib
