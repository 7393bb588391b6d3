// Hardware probe: issue rate of the four matrix-core instructions this library is built on, in
// CYCLES per instruction per SIMD (s_memtime, tick = shader cycle) and as chip-level T op/s at the
// clock the chip sustains while doing nothing else -- the register-only ceiling each MFMA kernel's
// roofline is priced against.  In particular SURVEY.md 8(d) asks for the fp64 peak (78.6 TFLOP/s is
// a datasheet figure that is not in the local guides) to be confirmed by a microbenchmark:
// v_mfma_f64_16x16x4_f64 = 2*16*16*4 = 2048 flop; 78.6 TF at 2.4 GHz over 1024 SIMDs = 32 flop/clk/SIMD
// <=> 64 cycles per instruction.
// One workgroup per CU with WPS waves per SIMD, NACC independent accumulators per wave, operands
// never change (register-only: no LDS, no memory).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 65536
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;
using i32x16 = __attribute__((ext_vector_type(16))) int;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x4 = __attribute__((ext_vector_type(4))) int;

// MODE 0: f32 32x32x2   1: f64 16x16x4   2: f16 32x32x16   3: i8 32x32x32
// MODE 4 / 5: f16 / i8 again, but every MFMA of the unrolled body gets its OWN pseudo-random operand
// registers (values in [1,10) / full-range bytes), so consecutive MFMAs toggle the multiplier inputs
// the way a GEMM on the reference's data does: the register-only POWER ceiling of the matrix cores on
// realistic operands (the constant-operand modes above measure the issue rate at an unrealistically
// low power).
template <int MODE, int NACC>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, float seed) {
  // only the state of THIS mode exists (everything else would spill: 512 threads = 256 registers each)
  constexpr bool F = MODE == 0 || MODE == 2 || MODE == 4, D = MODE == 1, I = MODE == 3 || MODE == 5;
  f32x16 af[F ? NACC : 1]; f64x4 ad[D ? NACC : 1]; i32x16 ai[I ? NACC : 1];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    if (F) af[i] = (f32x16)(seed * i);
    if (D) ad[i] = (f64x4)((double)seed * i);
    if (I) ai[i] = (i32x16)(i);
  }
  const float xf = seed + threadIdx.x; const double xd = seed + threadIdx.x;
  h8 xh;
#pragma unroll
  for (int j = 0; j < 8; ++j) xh[j] = (_Float16)(seed + j);
  i32x4 xi = {(int)threadIdx.x, 1, 2, 3};
  h8 rh[MODE == 4 ? 2 * NACC : 1];
  i32x4 ri[MODE == 5 ? 2 * NACC : 1];
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  if (MODE == 4 || MODE == 5) {
#pragma unroll
    for (int i = 0; i < 2 * NACC; ++i) {
      if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { st = st * 1664525u + 1013904223u; rh[i][j] = (_Float16)(1.0f + 9.0f * (st >> 8) * (1.0f / 16777216.0f)); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { st = st * 1664525u + 1013904223u; ri[i][j] = (int)st; }
      }
    }
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (MODE == 0) af[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, xf, af[i], 0, 0, 0);
      if constexpr (MODE == 1) ad[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, xd, ad[i], 0, 0, 0);
      if constexpr (MODE == 2) af[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, xh, af[i], 0, 0, 0);
      if constexpr (MODE == 3) ai[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xi, xi, ai[i], 0, 0, 0);
      if constexpr (MODE == 4) af[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh[2 * i], rh[2 * i + 1], af[i], 0, 0, 0);
      if constexpr (MODE == 5) ai[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ri[2 * i], ri[2 * i + 1], ai[i], 0, 0, 0);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    if (F) s += af[i][0];
    if (D) s += (float)ad[i][0];
    if (I) s += (float)ai[i][0];
  }
  asm volatile("s_nop 0" ::"v"(s) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int NACC>
int run(const char *name, double ops_per_inst, int wps, float *d, unsigned long long *dc) {
  const int blocks = 256, threads = wps * 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<MODE, NACC><<<blocks, threads>>>(d, dc, 0.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<MODE, NACC><<<blocks, threads>>>(d, dc, 0.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> c(blocks * 16);
  CHECK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
  // the SIMD's arbiter is not fair (priority, then age): co-resident waves finish at different times,
  // so the SIMD's busy time is the SLOWEST wave's elapsed, not the mean (a mean would under-count)
  double sum = 0; unsigned long long mx = 0; int cnt = 0;
  for (int b = 0; b < blocks; ++b) {
    unsigned long long bm = 0;
    for (int w = 0; w < wps * 4; ++w) bm = std::max(bm, c[b * 16 + w]);
    sum += bm; mx = std::max(mx, bm); ++cnt;
  }
  const double inst = (double)ITERS * NACC;
  const double cyc_per_inst = (sum / cnt) / (inst * wps);
  const double clock_ghz = (double)mx / (ms * 1e6);
  const double wall_tops = 256.0 * 4 * wps * inst * ops_per_inst / (ms * 1e-3) / 1e12;
  printf("%-26s acc %d wps %d  %7.2f cyc/instr/SIMD  clock >= %5.3f GHz  %8.1f T op/s wall  (%8.1f at 2.4 GHz and this cyc/instr)\n",
         name, NACC, wps, cyc_per_inst, clock_ghz, wall_tops, ops_per_inst / cyc_per_inst * 1024 * 2.4 / 1e3);
  return 0;
}

int main() {
  float *d; unsigned long long *dc;
  CHECK(hipMalloc(&d, 256 * 1024 * 4)); CHECK(hipMalloc(&dc, 256 * 16 * 8));
  for (int wps : {1, 2}) {
    run<1, 4>("v_mfma_f64_16x16x4_f64", 2048, wps, d, dc);
    run<1, 8>("v_mfma_f64_16x16x4_f64", 2048, wps, d, dc);
    run<0, 4>("v_mfma_f32_32x32x2_f32", 4096, wps, d, dc);
    run<2, 4>("v_mfma_f32_32x32x16_f16", 32768, wps, d, dc);
    run<3, 4>("v_mfma_i32_32x32x32_i8", 65536, wps, d, dc);
    run<4, 8>("f16 32x32x16, random operands", 32768, wps, d, dc);
    run<5, 8>("i8 32x32x32, random operands", 65536, wps, d, dc);
  }
  return 0;
}
