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
#define ITERS 16384
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;
using i32x16 = __attribute__((ext_vector_type(16))) int;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x4 = __attribute__((ext_vector_type(4))) int;

// MODE 0: f32 32x32x2   1: f64 16x16x4   2: f16 32x32x16   3: i8 32x32x32
template <int MODE, int NACC>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, float seed) {
  f32x16 af[NACC]; f64x4 ad[NACC]; i32x16 ai[NACC];
  for (int i = 0; i < NACC; ++i) { af[i] = (f32x16)(seed * i); ad[i] = (f64x4)((double)seed * i); ai[i] = (i32x16)(i); }
  const float xf = seed + threadIdx.x; const double xd = seed + threadIdx.x;
  h8 xh; for (int j = 0; j < 8; ++j) xh[j] = (_Float16)(seed + j);
  i32x4 xi = {(int)threadIdx.x, 1, 2, 3};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0) af[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, xf, af[i], 0, 0, 0);
      if (MODE == 1) ad[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, xd, ad[i], 0, 0, 0);
      if (MODE == 2) af[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, xh, af[i], 0, 0, 0);
      if (MODE == 3) ai[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xi, xi, ai[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += af[i][0] + (float)ad[i][0] + (float)ai[i][0];
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
  }
  return 0;
}
