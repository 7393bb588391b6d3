// Hardware probe: sustained issue rate of the VALU instructions a min-plus inner loop can be built
// from (wave64 on gfx950).  Every variant runs 16 independent accumulators per lane so that only
// issue rate matters.  Output: T lane-instructions per second and cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float x, float y) {
  float a[16];
  f2 p[8];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
  for (int i = 0; i < 8; ++i) p[i] = f2{(float)threadIdx.x, (float)i};
  const f2 xy = {x, y};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 2) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(xy));
      if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i & 7]) : "v"(xy));
      if (MODE == 6) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(x));
      if (MODE == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, float *d) {
  const int blocks = 256 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 1.0f, 2.0f);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, 1.0f, 2.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double lane_instr = (double)blocks * 256 * ITERS * 16;
  const double rate = lane_instr / (ms * 1e-3);
  printf("%-28s %7.2f T lane-instr/s   %5.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", name, rate / 1e12,
         1024 * 2.4e9 / (rate / 64));
}
int main() {
  float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_f32 (2 src)", d); run<1>("v_min_f32 (2 src)", d); run<2>("v_min3_f32 (3 distinct src)", d);
  run<6>("v_min3_f32 (2 distinct src)", d); run<3>("v_fma_f32", d); run<4>("v_pk_add_f32", d);
  run<5>("v_pk_fma_f32", d); run<7>("v_add_u32", d);
  return 0;
}
