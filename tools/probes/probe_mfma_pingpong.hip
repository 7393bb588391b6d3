// Hardware probe: how full ONE wavefront can keep its SIMD's matrix pipe with v_mfma_f32_16x16x32_f16 (16 cycles of pipe
// per instruction), and what the ping-pong organisation of the half / int8 kernels can reach at best -- registers only,
// no LDS, no memory, pseudo-random [1,10) operands, the kernel's own 8 x 4 block and MFMA order.
//   mode 0  two wavefronts per SIMD, both issuing MFMAs all the time (what probe_mfma_power measures: 16.1 cycles)
//   mode 1  ONE wavefront per SIMD issuing back to back
//   mode 2  two wavefronts per SIMD in the kernels' ping-pong: 32 MFMAs, s_barrier, (the partner's 32 MFMAs), s_barrier,
//           the two groups of four wavefronts one barrier apart, s_setprio 1 around the MFMAs -- the shipped kernels'
//           segment structure with EMPTY load segments: the ceiling of MfmaUtil for that structure
//   mode 3  mode 2 without the priority flips
//   mode 4  mode 2 with 64 MFMAs per segment (two slabs per segment)
// Prints cycles per MFMA per SIMD (16.0 = a full pipe) and the utilisation 16 / that.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ h8 rnd8(unsigned &st) {
  h8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) { st = st * 1664525u + 1013904223u; v[j] = (_Float16)(1.0f + 9.0f * (st >> 8) * (1.0f / 16777216.0f)); }
  return v;
}

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out, unsigned long long *cyc, int iters) {
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
  auto burst = [&](int q) {
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
  };
  auto sync = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (MODE <= 1) {
    for (int it = 0; it < iters; ++it) { burst(0); burst(1); }
  } else {
    const bool shifted = (threadIdx.x >> 8) == 1;     // waves 4-7: the SIMD partners of waves 0-3
    if (shifted) sync();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        sync();                                        // (end of this wavefront's empty load segment)
        if constexpr (MODE != 3) __builtin_amdgcn_s_setprio(1);
        burst(q);
        if constexpr (MODE == 4) burst(q ^ 1);
        if constexpr (MODE != 3) __builtin_amdgcn_s_setprio(0);
        sync();
      }
    }
    if (!shifted) sync();
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) s += acc[i][j][0];
  asm volatile("s_nop 0" ::"v"(s) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int THREADS>
int run(const char *name, float *d, unsigned long long *dc) {
  const int blocks = 256, iters = 1 << 15;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipMemset(dc, 0, 256 * 8 * 8));
  k<MODE, THREADS><<<blocks, THREADS>>>(d, dc, iters);   // warm-up: lets the power management settle on this load
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<MODE, THREADS><<<blocks, THREADS>>>(d, dc, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> c(blocks * 8);
  CHECK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long mx = 0;
  for (auto v : c) mx = std::max(mx, v);
  const double per_wave = (double)iters * 64.0 * (MODE == 4 ? 2.0 : 1.0);
  const double waves_per_simd = THREADS / 256.0;
  const double cyc_per_mfma_simd = (double)mx / (per_wave * waves_per_simd);
  const double tops = 256.0 * (THREADS / 64) * per_wave * 16384.0 / (ms * 1e-3) / 1e12;
  printf("%-78s %6.2f cyc/MFMA/SIMD  pipe %5.1f %%  %7.1f TF  clock >= %5.3f GHz\n", name, cyc_per_mfma_simd,
         100.0 * 16.0 / cyc_per_mfma_simd, tops, (double)mx / (ms * 1e6));
  return 0;
}

int main() {
  float *d; unsigned long long *dc;
  CHECK(hipMalloc(&d, 256 * 512 * 4)); CHECK(hipMalloc(&dc, 256 * 8 * 8));
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 512>("two wavefronts per SIMD, both issuing all the time", d, dc);
    run<1, 256>("ONE wavefront per SIMD, back to back", d, dc);
    run<2, 512>("ping-pong: 32 MFMAs | barrier | partner's 32 | barrier, priority flips (the kernels' structure)", d, dc);
    run<3, 512>("ping-pong without priority flips", d, dc);
    run<4, 512>("ping-pong with 64 MFMAs per segment", d, dc);
  }
  return 0;
}
