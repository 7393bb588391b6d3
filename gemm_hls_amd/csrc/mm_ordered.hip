// MM_PATH_ORDERED ("hw_emu"): the plain gfx950 kernel that evaluates each output element the way
// the reference's Naive does (include/Utility.h:18-42):
//     acc = Reduce::identity(); for k = 0..K-1: acc = Reduce(acc, Map(A[n,k], B[k,m]))
// one accumulator, k ascending, multiply and add as two separately rounded operations (this file
// is compiled with -ffp-contract=off), binary16 accumulating in binary16.  Bit-identical to the
// reference for every dtype and every (map, reduce); fully predicated, any N, K, M.
// LDS-tiled (64 x 64 outputs per 256-thread workgroup, 4 x 4 per thread) so that it is usable
// for verification at real sizes, but it is the parity anchor, not the fast path.
#include "mm_common.h"

namespace mm {
namespace {

constexpr int kTile = 64;   // outputs per workgroup edge
constexpr int kBK = 16;     // k-slab staged through LDS
constexpr int kPerThread = 4;

// ACC is the accumulator type: T itself for the Naive contract; float for the one exception,
// half (Multiply, Add) under MM_PATH_AUTO on shapes the matrix-core kernel does not take
// ("ordered_wide_f16": exact products, f32 accumulation, ONE rounding to binary16 on store -- the
// same contract as mfma_f16, so the AUTO path's half semantics do not change with the shape).
template <typename T, int MAP, int RED, bool AT, typename ACC = T>
__global__ __launch_bounds__(256) void ordered_kernel(const T *__restrict__ A, const T *__restrict__ B,
                                                      T *__restrict__ C, unsigned N, unsigned K,
                                                      unsigned M) {
  __shared__ T As[kBK][kTile + 1];  // [k][row], +1: column reads of a row-major source
  __shared__ T Bs[kBK][kTile];      // [k][col]
  const unsigned tid = threadIdx.x;
  const unsigned tx = tid % 16, ty = tid / 16;
  const unsigned row0 = blockIdx.y * kTile, col0 = blockIdx.x * kTile;

  ACC acc[kPerThread][kPerThread];
#pragma unroll
  for (int i = 0; i < kPerThread; ++i)
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) acc[i][j] = Op<RED, ACC>::identity();

  for (unsigned k0 = 0; k0 < K; k0 += kBK) {
    // stage A: 64 rows x 16 k
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned r, kk;
      if (AT) { r = tid % 64; kk = tid / 64 + 4 * i; }   // A is K x N: consecutive lanes along N
      else    { kk = tid % 16; r = tid / 16 + 16 * i; }  // A is N x K: consecutive lanes along K
      const unsigned gr = row0 + r, gk = k0 + kk;
      T v = (T)0;
      if (gr < N && gk < K) v = AT ? A[(size_t)gk * N + gr] : A[(size_t)gr * K + gk];
      As[kk][r] = v;
    }
    // stage B: 16 k x 64 cols
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned c = tid % 64, kk = tid / 64 + 4 * i;
      const unsigned gc = col0 + c, gk = k0 + kk;
      Bs[kk][c] = (gc < M && gk < K) ? B[(size_t)gk * M + gc] : (T)0;
    }
    __syncthreads();
    const unsigned kmax = (K - k0) < (unsigned)kBK ? (K - k0) : (unsigned)kBK;
    for (unsigned kk = 0; kk < kmax; ++kk) {  // strictly ascending k
      T av[kPerThread], bv[kPerThread];
#pragma unroll
      for (int i = 0; i < kPerThread; ++i) av[i] = As[kk][ty * kPerThread + i];
#pragma unroll
      for (int j = 0; j < kPerThread; ++j) bv[j] = Bs[kk][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < kPerThread; ++i)
#pragma unroll
        for (int j = 0; j < kPerThread; ++j)
          acc[i][j] = Op<RED, ACC>::apply(acc[i][j], Op<MAP, ACC>::apply((ACC)av[i], (ACC)bv[j]));
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < kPerThread; ++i) {
    const unsigned gr = row0 + ty * kPerThread + i;
    if (gr >= N) continue;
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
      const unsigned gc = col0 + tx + 16 * j;
      if (gc < M) C[(size_t)gr * M + gc] = (T)acc[i][j];
    }
  }
}

template <typename T, int MAP, int RED>
int launch_t(hipStream_t s, const Problem &p) {
  if (p.n == 0 || p.m == 0) return 0;
  dim3 grid((p.m + kTile - 1) / kTile, (p.n + kTile - 1) / kTile);
  if (p.a_transposed)
    hipLaunchKernelGGL((ordered_kernel<T, MAP, RED, true>), grid, dim3(256), 0, s, (const T *)p.a,
                       (const T *)p.b, (T *)p.c, p.n, p.k, p.m);
  else
    hipLaunchKernelGGL((ordered_kernel<T, MAP, RED, false>), grid, dim3(256), 0, s, (const T *)p.a,
                       (const T *)p.b, (T *)p.c, p.n, p.k, p.m);
  return (int)hipGetLastError();
}

template <typename T, int MAP>
int launch_red(hipStream_t s, int red, const Problem &p) {
  switch (red) {
    case MM_OP_ADD: return launch_t<T, MAP, MM_OP_ADD>(s, p);
    case MM_OP_MULTIPLY: return launch_t<T, MAP, MM_OP_MULTIPLY>(s, p);
    case MM_OP_AND: return launch_t<T, MAP, MM_OP_AND>(s, p);
    case MM_OP_MIN: return launch_t<T, MAP, MM_OP_MIN>(s, p);
    case MM_OP_MAX: return launch_t<T, MAP, MM_OP_MAX>(s, p);
  }
  return kErrNotSupported;
}

template <typename T>
int launch_map(hipStream_t s, const mm_config_t &cfg, const Problem &p) {
  switch (cfg.map_op) {
    case MM_OP_ADD: return launch_red<T, MM_OP_ADD>(s, cfg.reduce_op, p);
    case MM_OP_MULTIPLY: return launch_red<T, MM_OP_MULTIPLY>(s, cfg.reduce_op, p);
    case MM_OP_AND: return launch_red<T, MM_OP_AND>(s, cfg.reduce_op, p);
    case MM_OP_MIN: return launch_red<T, MM_OP_MIN>(s, cfg.reduce_op, p);
    case MM_OP_MAX: return launch_red<T, MM_OP_MAX>(s, cfg.reduce_op, p);
  }
  return kErrNotSupported;
}

}  // namespace

int launch_half_wide(hipStream_t s, const Problem &p) {
  if (p.n == 0 || p.m == 0) return 0;
  dim3 grid((p.m + kTile - 1) / kTile, (p.n + kTile - 1) / kTile);
  if (p.a_transposed)
    hipLaunchKernelGGL((ordered_kernel<half_t, MM_OP_MULTIPLY, MM_OP_ADD, true, float>), grid, dim3(256), 0, s,
                       (const half_t *)p.a, (const half_t *)p.b, (half_t *)p.c, p.n, p.k, p.m);
  else
    hipLaunchKernelGGL((ordered_kernel<half_t, MM_OP_MULTIPLY, MM_OP_ADD, false, float>), grid, dim3(256), 0, s,
                       (const half_t *)p.a, (const half_t *)p.b, (half_t *)p.c, p.n, p.k, p.m);
  return (int)hipGetLastError();
}

int launch_ordered(hipStream_t s, const mm_config_t &cfg, const Problem &p) {
  switch (cfg.dtype) {
    case MM_DTYPE_F32: return launch_map<float>(s, cfg, p);
    case MM_DTYPE_F64: return launch_map<double>(s, cfg, p);
    case MM_DTYPE_F16: return launch_map<half_t>(s, cfg, p);
    case MM_DTYPE_I8: return launch_map<int8_t>(s, cfg, p);
    case MM_DTYPE_U8: return launch_map<uint8_t>(s, cfg, p);
    case MM_DTYPE_I16: return launch_map<int16_t>(s, cfg, p);
    case MM_DTYPE_U16: return launch_map<uint16_t>(s, cfg, p);
    case MM_DTYPE_I32: return launch_map<int32_t>(s, cfg, p);
    case MM_DTYPE_U32: return launch_map<uint32_t>(s, cfg, p);
    case MM_DTYPE_I64: return launch_map<int64_t>(s, cfg, p);
    case MM_DTYPE_U64: return launch_map<uint64_t>(s, cfg, p);
  }
  return kErrNotSupported;
}

}  // namespace mm
