// K x N -> N x K transposition pre-pass for 1- and 2-byte element types (MM_TRANSPOSED_A: the reference's
// ReadATransposed layout, kernel/Memory.cpp:205-261).  The half / int8 matrix-core kernels gather a K x N A slab out of
// LDS with twice the transpose-read instructions of a row-major one (24 instead of 16 LDS reads per wavefront and slab:
// every operand of v_mfma_*_16x16x* wants consecutive k of one row) and run 6-10 % behind the row-major kernels at every
// size.  The pre-pass moves N x K elements once in and once out -- O(N K) against the product's O(N K M) -- so for wide
// enough M it costs less than that: measured 3.5 % of the product at M = 16384, 2.2 % at 32768, 5 % at 6144-8192 for
// half (uint8: 3.4 / 2.1 / 8-11 %), against the K x N kernels' 7-15 % (profiles/r04b_kxn_prepass.txt,
// r04c_kxn_prepass_forced_small_m.txt); at 4096 the pass costs more than it returns.
// The product then runs the row-major default and gives ITS bits: a K x N call equals the row-major call.
//
// dst[n][k] = src[k][n], 128-byte x 128-byte element tiles through LDS: 16-byte loads along n (one request per line),
// element gather from LDS, 16-byte stores along k (one request per line).  Edges are predicated per 16-byte chunk
// (N and K are multiples of the chunk: 8 halves / 16 bytes -- guaranteed by the callers' serves() rules).
#include "mm_common.h"

namespace mm {
namespace {

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

template <typename T>
__global__ __launch_bounds__(256) void transpose_kxn_small_kernel(const T *__restrict__ src, T *__restrict__ dst, unsigned K, unsigned N) {
  constexpr unsigned ES = sizeof(T), EPC = 16 / ES, TS = 128 / ES;   // elements per 16-byte chunk; tile side (128 bytes)
  // LDS row pitch 132 B: a gathering wavefront reads 8 output chunks (k blocks EPC rows apart) x 8 columns per instruction;
  // EPC rows of 33 words put the k blocks 8 banks apart for 2-byte elements (conflict-free) and 16 apart for bytes (2-way)
  constexpr unsigned PITCH = 128 + 4;
  __shared__ __attribute__((aligned(16))) char tile[TS * PITCH];
  const unsigned blocks_n = (N + TS - 1) / TS;   // 1-D grid: either dimension may exceed the 65535 of gridDim.y
  const unsigned k0 = (blockIdx.x / blocks_n) * TS, n0 = (blockIdx.x % blocks_n) * TS;
  constexpr unsigned CHUNKS = TS * 8;             // 16-byte chunks in the tile
#pragma unroll
  for (unsigned i = 0; i < CHUNKS / 256; ++i) {
    const unsigned c = threadIdx.x + 256 * i, kr = c >> 3, ch = c & 7u;
    u32x4 v = (u32x4)0u;
    if (k0 + kr < K && n0 + ch * EPC < N) v = *(const u32x4 *)(src + (size_t)(k0 + kr) * N + n0 + ch * EPC);
#pragma unroll
    for (int j = 0; j < 4; ++j) *(unsigned *)(tile + kr * PITCH + ch * 16 + 4 * j) = v[j];   // rows are only 4-byte aligned
  }
  __syncthreads();
#pragma unroll
  for (unsigned i = 0; i < CHUNKS / 256; ++i) {
    const unsigned c = threadIdx.x + 256 * i, kc = c & 7u, n = c >> 3;   // 8 lanes = the 128 bytes of one output row
    union { T e[EPC]; u32x4 v; } out;
#pragma unroll
    for (unsigned j = 0; j < EPC; ++j) out.e[j] = *(const T *)(tile + (kc * EPC + j) * PITCH + n * ES);
    if (n0 + n < N && k0 + kc * EPC < K) *(u32x4 *)(dst + (size_t)(n0 + n) * K + k0 + kc * EPC) = out.v;
  }
}

}  // namespace

int launch_transpose_kxn(hipStream_t s, const void *src, void *dst, unsigned K, unsigned N, unsigned elem_size) {
  (void)hipGetLastError();
  if (elem_size == 2) {
    hipLaunchKernelGGL(transpose_kxn_small_kernel<unsigned short>, dim3(((N + 63) / 64) * ((K + 63) / 64)), dim3(256), 0, s,
                       (const unsigned short *)src, (unsigned short *)dst, K, N);
  } else if (elem_size == 1) {
    hipLaunchKernelGGL(transpose_kxn_small_kernel<unsigned char>, dim3(((N + 127) / 128) * ((K + 127) / 128)), dim3(256), 0, s,
                       (const unsigned char *)src, (unsigned char *)dst, K, N);
  } else {
    return kErrNotSupported;
  }
  return (int)hipGetLastError();
}

// K x N A through the pre-pass: worth it where the row-major default kernel serves the transposed problem and M is wide
// enough for the O(N K) pass to cost less than the K x N kernel's deficit: from M = 6144 (kxn_prepass_min_m overrides).
bool transposes_first_small(const Problem &p, unsigned elem_size) {
  const int knob = tuning(TUNE_KXN_PREPASS_MIN_M);
  const unsigned min_m = knob >= 0 ? (unsigned)knob : 6144u;   // measured: profiles/r04c_kxn_prepass_forced_small_m.txt
  return p.a_transposed && p.m >= min_m && p.n >= 1024 && (unsigned long long)p.n * p.k * elem_size <= (4ull << 30);
}

}  // namespace mm
