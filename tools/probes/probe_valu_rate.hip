// Hardware probe: sustained issue rate of the VALU instructions a min-plus inner loop can be built
// from (wave64 on gfx950) -- the ceiling the valu_tile kernel is priced against.
//
// Round-1's version assumed a 2.4 GHz clock when converting wall time into cycles; under a pure
// VALU load the chip does not hold 2.4 GHz, so its "cycles per instruction" were too high and the
// kernel appeared to beat its own ceiling.  This version measures CYCLES
// directly: every wave brackets its instruction stream with s_memtime (tick = shader cycle,
// MI355X_MICROARCH.md constants table), exactly one workgroup of WPS*4 waves runs per CU (WPS waves
// per SIMD, all co-resident for the whole measurement), and
//     cycles per wave-instruction per SIMD = slowest wave's elapsed / (instructions per wave * WPS)
// (slowest, not mean: the SIMD arbiter favours the older wave, so co-resident waves finish apart).
// The delivered clock = (cycles the slowest wave counted) / (wall time of the launch) is reported
// next to it, so a ceiling in op/s can be stated at the clock the chip actually sustains.
// 16 independent accumulators per lane: dependency latency never limits issue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define ITERS 32768
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, float x, float y) {
  float a[16];
  f2 p[8];
  double da[16];   // fp64 modes (round 3: the DMA-staged VALU kernel serves double min-plus)
  for (int i = 0; i < 16; ++i) da[i] = threadIdx.x + i;
  const double dx = x, dy = y;
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  for (int i = 0; i < 8; ++i) p[i] = f2{(float)threadIdx.x, (float)i};
  const f2 xy = {x, y};
  h2 hp[16];       // round 6: the unfused multiply-add steps of the k-ordered tile kernel (mm_valu_tile_fp_exact.hip)
  for (int i = 0; i < 16; ++i) hp[i] = h2{(_Float16)(threadIdx.x & 7), (_Float16)i};
  const h2 hxy = {(_Float16)x, (_Float16)y};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 2) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(xy));
      if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i & 7]) : "v"(xy));
      if (MODE == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 8) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 11) asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[i]) : "v"(dx));
      if (MODE == 12) asm volatile("v_min_f64 %0, %0, %1" : "+v"(da[i]) : "v"(dx));
      if (MODE == 13) {   // the double min-plus step: add into a temporary, min into the accumulator (there is no v_min3_f64)
        double s0;
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(s0) : "v"(dx), "v"(da[(i + 1) & 15]));
        asm volatile("v_min_f64 %0, %0, %1" : "+v"(da[i]) : "v"(s0));
      }
      if (MODE == 14) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 15) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(xy));
      if (MODE == 16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(hp[i]) : "v"(hxy));
      if (MODE == 17) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(hp[i]) : "v"(hxy));
      if (MODE == 18) {   // float multiply-add, unfused: a product into a temporary, one add into the accumulator
        float s0;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(x), "v"(a[(i + 1) & 15]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s0));
      }
      if (MODE == 19) {   // the same packed: two elements per lane and instruction
        f2 s0;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(xy), "v"(p[(i + 1) & 7]));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(s0));
      }
      if (MODE == 20) {   // half: binary16 product, binary16 sum, two elements per lane and instruction
        h2 s0;
        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(s0) : "v"(hxy), "v"(hp[(i + 1) & 15]));
        asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(hp[i]) : "v"(s0));
      }
      if (MODE == 21) {   // double
        double s0;
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(s0) : "v"(dx), "v"(da[(i + 1) & 15]));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(da[i]) : "v"(s0));
      }
      // the float min-plus inner loop as valu_tile compiles it: per accumulator and pair of k-steps
      // two adds into temporaries and ONE v_min3 (acc, s0, s1): 3 instructions per 4 operations
      if (MODE == 9) {
        float s0, s1;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x), "v"(a[(i + 1) & 15]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(y), "v"(a[(i + 2) & 15]));
        asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s0), "v"(s1));
      }
      // the same with a plain 2-source min twice (what it would be without v_min3)
      if (MODE == 10) {
        float s0, s1;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x), "v"(a[(i + 1) & 15]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(y), "v"(a[(i + 2) & 15]));
        asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s0));
        asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s1));
      }
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  for (int i = 0; i < 16; ++i) s += (float)da[i];
  for (int i = 0; i < 16; ++i) s += (float)hp[i][0] + (float)hp[i][1];
  (void)dy;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
int run(const char *name, int insts_per_slot, double ops_per_slot, int wps, float *d, unsigned long long *dc) {
  const int blocks = 256, threads = wps * 256;  // one workgroup per CU, wps waves on each SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<MODE><<<blocks, threads>>>(d, dc, 1.0f, 2.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<MODE><<<blocks, threads>>>(d, dc, 1.0f, 2.0f);
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
  const double per_wave = sum / cnt;
  const double inst = (double)ITERS * 16 * insts_per_slot;
  const double cyc_per_inst = per_wave / (inst * wps);
  const double clock_ghz = (double)mx / (ms * 1e6);   // slowest wave's cycles over the launch's wall time (lower bound)
  const double ops_per_cyc_simd = 64.0 * ops_per_slot / (cyc_per_inst * insts_per_slot);
  printf("%-34s wps %d  %6.3f cyc/wave-instr/SIMD  clock >= %5.3f GHz  %7.2f T lane-instr/s wall", name, wps, cyc_per_inst,
         clock_ghz, 256.0 * threads * inst / (ms * 1e-3) / 1e12);
  if (ops_per_slot > 0)
    printf("  | %6.2f op/clk/SIMD -> %6.1f TOp/s at this clock, %6.1f at 2.4 GHz", ops_per_cyc_simd,
           ops_per_cyc_simd * 1024 * clock_ghz / 1e3, ops_per_cyc_simd * 1024 * 2.4 / 1e3);
  printf("\n");
  return 0;
}

int main() {
  float *d; unsigned long long *dc;
  CHECK(hipMalloc(&d, 256 * 1024 * 4)); CHECK(hipMalloc(&dc, 256 * 16 * 8));
  for (int wps : {1, 2, 4}) {  // 4 waves per SIMD = 1024 threads, the largest single workgroup
    run<0>("v_add_f32 (2 src)", 1, 0, wps, d, dc);
    run<1>("v_min_f32 (2 src)", 1, 0, wps, d, dc);
    run<6>("v_max_f32 (2 src)", 1, 0, wps, d, dc);
    run<2>("v_min3_f32 (3 src)", 1, 0, wps, d, dc);
    run<3>("v_fma_f32", 1, 0, wps, d, dc);
    run<4>("v_pk_add_f32", 1, 0, wps, d, dc);
    run<5>("v_pk_fma_f32", 1, 0, wps, d, dc);
    run<7>("v_add_u32", 1, 0, wps, d, dc);
    run<8>("v_min_u32", 1, 0, wps, d, dc);
    run<9>("min-plus mix: 2 v_add + 1 v_min3", 3, 4, wps, d, dc);
    run<10>("min-plus mix: 2 v_add + 2 v_min", 4, 4, wps, d, dc);
    run<11>("v_add_f64", 1, 0, wps, d, dc);
    run<12>("v_min_f64", 1, 0, wps, d, dc);
    run<13>("fp64 min-plus: v_add_f64 + v_min_f64", 2, 2, wps, d, dc);
    run<14>("v_mul_f32", 1, 0, wps, d, dc);
    run<15>("v_pk_mul_f32", 1, 0, wps, d, dc);
    run<16>("v_pk_mul_f16", 1, 0, wps, d, dc);
    run<17>("v_pk_add_f16", 1, 0, wps, d, dc);
    run<18>("unfused mul-add f32: v_mul + v_add", 2, 2, wps, d, dc);
    run<19>("unfused mul-add f32 packed: v_pk_mul + v_pk_add", 2, 4, wps, d, dc);
    run<20>("unfused mul-add f16 packed: v_pk_mul_f16 + v_pk_add_f16", 2, 4, wps, d, dc);
    run<21>("unfused mul-add f64: v_mul_f64 + v_add_f64", 2, 2, wps, d, dc);
  }
  return 0;
}
