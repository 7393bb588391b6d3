// Hardware probe: what the matrix cores sustain from REGISTERS ONLY (no LDS, no memory) on pseudo-random
// [1,10) binary16 operands, as a function of how the operand registers change from one MFMA to the next.
// The half GEMM kernel is power-limited, so the order in which a wavefront walks its
// 4 x 2 block of 32x32 accumulators -- which operand stays on the bus between consecutive MFMAs -- and the
// instruction shape (32x32x16 vs 16x16x32: operand bytes vs accumulator bytes per flop) are energy knobs.
//   mode 0  every MFMA has its own A and B registers (the ceiling round 2 quoted: nothing is reused)
//   mode 1  GEMM order of the shipped kernel: for ks, for mi (4), for ni (2): A stays for 2 MFMAs, B alternates
//   mode 2  serpentine: ni runs 0,1,1,0,...: exactly one operand changes between consecutive MFMAs
//   mode 3  B-stationary: for ni, for mi (serpentine in mi): B stays for 4 MFMAs
//   mode 4  v_mfma_f32_16x16x32_f16 over the same 128 x 64 wavefront tile (8 x 4 accumulators of 4 registers)
//   mode 5  3 x 4 block (the 96 x 128 wavefront tile of a 384 x 256 workgroup tile), serpentine
// Two operand sets (two k-steps) alternate, as in the kernel.  One 512-thread workgroup per CU (2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <type_traits>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ h8 rnd8(unsigned &st) {
  h8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) { st = st * 1664525u + 1013904223u; v[j] = (_Float16)(1.0f + 9.0f * (st >> 8) * (1.0f / 16777216.0f)); }
  return v;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  float s = 0;
  unsigned long long t0, t1;
  if constexpr (MODE == 4) {
    constexpr int TA = 8, TB = 4;
    h8 a[2][TA], b[2][TB];
    f32x4 acc[TA][TB];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int i = 0; i < TA; ++i) a[q][i] = rnd8(st);
#pragma unroll
      for (int i = 0; i < TB; ++i) b[q][i] = rnd8(st);
    }
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4)0.0f;
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
          for (int jj = 0; jj < TB; ++jj) {
            const int j = (i & 1) ? TB - 1 - jj : jj;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
          }
    }
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) s += acc[i][j][0];
    asm volatile("s_nop 0" ::"v"(s) : "memory");
    t1 = __builtin_amdgcn_s_memtime();
  } else {
    constexpr int TA = MODE == 5 ? 3 : 4, TB = MODE == 5 ? 4 : 2, NM = TA * TB;
    constexpr int NQ = MODE == 0 ? 1 : 2;  // mode 0: one set of 8 + 8 operand registers (two would spill)
    h8 a[NQ][MODE == 0 ? NM : TA], b[NQ][MODE == 0 ? NM : TB];
    f32x16 acc[TA][TB];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
      for (int i = 0; i < (MODE == 0 ? NM : TA); ++i) a[q][i] = rnd8(st);
#pragma unroll
      for (int i = 0; i < (MODE == 0 ? NM : TB); ++i) b[q][i] = rnd8(st);
    }
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = (f32x16)0.0f;
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if constexpr (MODE == 3) {
#pragma unroll
          for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int ii = 0; ii < TA; ++ii) {
              const int i = (j & 1) ? TA - 1 - ii : ii;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
          for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int jj = 0; jj < TB; ++jj) {
              const int j = ((MODE == 2 || MODE == 5) && (i & 1)) ? TB - 1 - jj : jj;
              if constexpr (MODE == 0)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i * TB + jj], b[0][i * TB + jj], acc[i][j], 0, 0, 0);
              else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
            }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) s += acc[i][j][0];
    asm volatile("s_nop 0" ::"v"(s) : "memory");
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}


// The same comparison for the int8 and bf16 matrix-core instructions (full-range random bytes / random bf16 in [1,10)):
//   KIND 0: i32_32x32x32_i8, 4x2 block   1: i32_16x16x64_i8, 8x4 block   2: f32_32x32x16_bf16, 4x2   3: f32_16x16x32_bf16, 8x4
using i32x16 = __attribute__((ext_vector_type(16))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using b8 = __attribute__((ext_vector_type(8))) __bf16;
template <int KIND>
__global__ __launch_bounds__(512) void k2(float *out, unsigned long long *cyc, int iters) {
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  constexpr bool BIG = KIND == 0 || KIND == 2, INT = KIND < 2;
  constexpr int TA = BIG ? 4 : 8, TB = BIG ? 2 : 4;
  using op_t = typename std::conditional<INT, i32x4, b8>::type;
  using acc_t = typename std::conditional<INT, typename std::conditional<BIG, i32x16, i32x4>::type,
                                          typename std::conditional<BIG, f32x16, f32x4>::type>::type;
  op_t a[2][TA], b[2][TB];
  acc_t acc[TA][TB];
  auto rnd = [&]() {
    op_t v;
    if constexpr (INT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { st = st * 1664525u + 1013904223u; v[j] = (int)st; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { st = st * 1664525u + 1013904223u; v[j] = (__bf16)(1.0f + 9.0f * (st >> 8) * (1.0f / 16777216.0f)); }
    }
    return v;
  };
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int i = 0; i < TA; ++i) a[q][i] = rnd();
#pragma unroll
    for (int i = 0; i < TB; ++i) b[q][i] = rnd();
  }
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = (acc_t)0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          if constexpr (KIND == 0) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
          if constexpr (KIND == 1) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
          if constexpr (KIND == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
          if constexpr (KIND == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
        }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) s += (float)acc[i][j][0];
  asm volatile("s_nop 0" ::"v"(s) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// fp32 / fp64 matrix instructions, the same question: v_mfma_f32_32x32x2_f32 (64 cycles) vs v_mfma_f32_16x16x4_f32 (32 cycles),
// random operands, 64 x 128 wavefront tile (2 x 4 accumulators of 16 registers, or 4 x 8 of 4).
//   KIND 0: f32 32x32x2, 2x4 block   1: f32 16x16x4, 4x8 block
template <int KIND>
__global__ __launch_bounds__(512) void k3(float *out, unsigned long long *cyc, int iters) {
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  constexpr bool BIG = KIND == 0;
  constexpr int TA = BIG ? 2 : 4, TB = BIG ? 4 : 8, KS = 4;   // KS k-steps' worth of distinct operand registers
  using acc_t = typename std::conditional<BIG, f32x16, f32x4>::type;
  float a[KS][TA], b[KS][TB];
  acc_t acc[TA][TB];
#pragma unroll
  for (int q = 0; q < KS; ++q) {
#pragma unroll
    for (int i = 0; i < TA; ++i) { st = st * 1664525u + 1013904223u; a[q][i] = 1.0f + 9.0f * (st >> 8) * (1.0f / 16777216.0f); }
#pragma unroll
    for (int i = 0; i < TB; ++i) { st = st * 1664525u + 1013904223u; b[q][i] = 1.0f + 9.0f * (st >> 8) * (1.0f / 16777216.0f); }
  }
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = (acc_t)0.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < KS; ++q)
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          if constexpr (BIG) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
        }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) s += acc[i][j][0];
  asm volatile("s_nop 0" ::"v"(s) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// Accumulators in AGPRs vs VGPRs (unified file on gfx950, but the operand paths differ): 16x16x32 f16, 8x4 block, the MFMAs
// written in inline asm with the accumulator constrained to the accumulation registers ("a") or the vector registers ("v").
template <bool AGPR>
__global__ __launch_bounds__(512) void k4(float *out, unsigned long long *cyc, int iters) {
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  constexpr int TA = 8, TB = 4;
  h8 a[2][TA], b[2][TB];
  f32x4 acc[TA][TB];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int i = 0; i < TA; ++i) a[q][i] = rnd8(st);
#pragma unroll
    for (int i = 0; i < TB; ++i) b[q][i] = rnd8(st);
  }
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4)0.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[q][i]), "v"(b[q][j]));
          else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[q][i]), "v"(b[q][j]));
        }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) s += acc[i][j][0];
  asm volatile("s_nop 0" ::"v"(s) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// The K = 16 form of the 16x16 bf16 instruction (v_mfma_f32_16x16x16_bf16, 4 bf16 per lane and operand): 8x4 block, the
// candidate for MM_PATH_SPLIT's kernel, whose 16-deep packed slabs it could consume without a new layout.
typedef short s4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k5(float *out, unsigned long long *cyc, int iters) {
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  constexpr int TA = 8, TB = 4;
  s4v a[2][TA], b[2][TB];
  f32x4 acc[TA][TB];
  auto rnd = [&]() {
    s4v v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st = st * 1664525u + 1013904223u;
      const float f = 1.0f + 9.0f * (st >> 8) * (1.0f / 16777216.0f);
      v[j] = (short)(__builtin_bit_cast(unsigned, f) >> 16);
    }
    return v;
  };
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int i = 0; i < TA; ++i) a[q][i] = rnd();
#pragma unroll
    for (int i = 0; i < TB; ++i) b[q][i] = rnd();
  }
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4)0.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) s += acc[i][j][0];
  asm volatile("s_nop 0" ::"v"(s) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
int run(const char *name, int mfma_per_iter, double flop_per_mfma, float *d, unsigned long long *dc) {
  const int blocks = 256, iters = 1 << 17;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto launch = [&]() {
    if constexpr (MODE == 40) k5<<<blocks, 512>>>(d, dc, iters);
    else if constexpr (MODE >= 30) k4<MODE == 31><<<blocks, 512>>>(d, dc, iters);
    else if constexpr (MODE >= 20) k3<MODE - 20><<<blocks, 512>>>(d, dc, iters / 8);
    else if constexpr (MODE >= 10) k2<MODE - 10><<<blocks, 512>>>(d, dc, iters);
    else k<MODE><<<blocks, 512>>>(d, dc, iters);
  };
  launch();   // warm-up: lets the power management settle on this load
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> c(blocks * 8);
  CHECK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long mx = 0;
  for (auto v : c) mx = std::max(mx, v);
  const double inst = (double)((MODE >= 20 && MODE < 30) ? iters / 8 : iters) * mfma_per_iter;   // per wave
  const double tops = 256.0 * 8 * inst * flop_per_mfma / (ms * 1e-3) / 1e12;
  printf("%-58s %7.1f TF  clock >= %5.3f GHz  %6.2f cyc/MFMA/SIMD  %7.2f ms\n", name, tops, (double)mx / (ms * 1e6),
         (double)mx / (inst * 2), ms);
  return 0;
}

int main() {
  float *d; unsigned long long *dc;
  CHECK(hipMalloc(&d, 256 * 512 * 4)); CHECK(hipMalloc(&dc, 256 * 8 * 8));
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("32x32x16, own A and B registers per MFMA", 16, 32768, d, dc);
    run<1>("32x32x16, 4x2 block, mi-major (shipped order)", 16, 32768, d, dc);
    run<2>("32x32x16, 4x2 block, serpentine", 16, 32768, d, dc);
    run<3>("32x32x16, 4x2 block, B-stationary serpentine", 16, 32768, d, dc);
    run<4>("16x16x32, 8x4 block, serpentine", 64, 16384, d, dc);
    run<5>("32x32x16, 3x4 block, serpentine", 24, 32768, d, dc);
    run<10>("i8 32x32x32, 4x2 block", 16, 65536, d, dc);
    run<11>("i8 16x16x64, 8x4 block", 64, 32768, d, dc);
    run<12>("bf16 32x32x16, 4x2 block", 16, 32768, d, dc);
    run<13>("bf16 16x16x32, 8x4 block", 64, 16384, d, dc);
    run<40>("bf16 16x16x16 (K = 16 form), 8x4 block", 64, 8192, d, dc);
    run<30>("f16 16x16x32 (asm), accumulators in VGPRs", 64, 16384, d, dc);
    run<31>("f16 16x16x32 (asm), accumulators in AGPRs", 64, 16384, d, dc);
    run<20>("f32 32x32x2, 2x4 block", 32, 4096, d, dc);
    run<21>("f32 16x16x4, 4x8 block", 128, 2048, d, dc);
  }
  return 0;
}
