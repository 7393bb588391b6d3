// Device-side synthetic input fill for timing-only runs: the reference's `verify off` mode leaves
// its device buffers uninitialised (host/RunHardware.cpp:99,140); here they get the same value
// DISTRIBUTION as the verified runs (uniform on [1,10): real for float/double/half, integer
// 1..10 for integral types, host/RunHardware.cpp:31-35) from a counter-based generator, so that
// clocks and power look like a real run.  Not the seed-5 host stream: verified runs copy that in.
#include "mm_common.h"

namespace mm {
namespace {

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <typename T, bool INTEGRAL>
__global__ void fill_kernel(T *p, size_t n, unsigned long long seed) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const unsigned long long r = splitmix64(seed ^ (i * 0xD1342543DE82EF95ull));
    if (INTEGRAL) {
      p[i] = (T)(1 + (r >> 33) % 10);
    } else {
      const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
      p[i] = (T)(1.0 + 9.0 * u);
    }
  }
}

template <typename T, bool INTEGRAL>
int fill_t(hipStream_t s, void *ptr, size_t n, unsigned long long seed) {
  if (n == 0) return 0;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL((fill_kernel<T, INTEGRAL>), dim3(blocks), dim3(256), 0, s, (T *)ptr, n, seed);
  return (int)hipGetLastError();
}

}  // namespace

int launch_fill(hipStream_t s, mm_dtype_t dtype, void *ptr, size_t n, unsigned long long seed) {
  switch (dtype) {
    case MM_DTYPE_F32: return fill_t<float, false>(s, ptr, n, seed);
    case MM_DTYPE_F64: return fill_t<double, false>(s, ptr, n, seed);
    case MM_DTYPE_F16: return fill_t<half_t, false>(s, ptr, n, seed);
    case MM_DTYPE_I8: return fill_t<int8_t, true>(s, ptr, n, seed);
    case MM_DTYPE_U8: return fill_t<uint8_t, true>(s, ptr, n, seed);
    case MM_DTYPE_I16: return fill_t<int16_t, true>(s, ptr, n, seed);
    case MM_DTYPE_U16: return fill_t<uint16_t, true>(s, ptr, n, seed);
    case MM_DTYPE_I32: return fill_t<int32_t, true>(s, ptr, n, seed);
    case MM_DTYPE_U32: return fill_t<uint32_t, true>(s, ptr, n, seed);
    case MM_DTYPE_I64: return fill_t<int64_t, true>(s, ptr, n, seed);
    case MM_DTYPE_U64: return fill_t<uint64_t, true>(s, ptr, n, seed);
  }
  return kErrNotSupported;
}

}  // namespace mm
