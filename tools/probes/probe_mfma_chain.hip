// Hardware probe: does v_mfma_f32_32x32x2_f32 sustain full rate when consecutive instructions of a wavefront all
// accumulate into the SAME 32x32 accumulator (a 32 x 32 wavefront tile: one dependent chain), and how many wavefronts per
// SIMD does it take to hide the dependency if not?  NACC accumulators are used round-robin; WPS wavefronts per SIMD.
// Output: TFLOP/s of the launch (wall clock) per (NACC, WPS).  Built by hand: hipcc --offload-arch=gfx950 -O3 -o probe ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITERS 4096
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(1024) void k(float *out, float x, float y) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x16)(float)threadIdx.x;
  float a = x + threadIdx.x, b = y;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % NACC], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
int run(int wps) {
  float *out;
  CHECK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int threads = 256 * wps;
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 1.0f, 2.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 1.0f, 2.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 256.0 * 4 * wps * ITERS * 16 * (32.0 * 32 * 2 * 2);
  printf("accumulators %d  wavefronts/SIMD %d  %.1f TFLOP/s\n", NACC, wps, flop / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int wps = 1; wps <= 2; ++wps) {
    if (run<1>(wps) || run<2>(wps) || run<4>(wps) || run<8>(wps)) return 1;
  }
  return 0;
}
