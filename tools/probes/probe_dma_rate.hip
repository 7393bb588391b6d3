// Hardware probe: how fast can one CU pull L2-resident data into LDS with global_load_lds_dwordx4
// (LDS-DMA), alone and next to ds_read_b128 traffic?  One 512-thread workgroup per CU; every wave
// issues DMA pieces of 1 KiB from a 64-KiB per-workgroup source window (L2-resident after the first
// pass) into a 64-KiB LDS ring.  Reports bytes per clock per CU at 2.4 GHz nominal.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef float f4 __attribute__((ext_vector_type(4)));
#define ITERS 2048
template <int MODE>  // 0: DMA only, 1: ds_read only, 2: both
__global__ __launch_bounds__(512) void k(const char *src, float *out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char *base = src + (size_t)blockIdx.x * 65536;
  f4 acc = {0, 0, 0, 0};
  for (int it = 0; it < ITERS; ++it) {
    const unsigned piece = (it * 8 + wave) & 63;  // 64 pieces of 1 KiB
    if (MODE != 1)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + piece * 1024 + lane * 16), (lptr_t)(smem + piece * 1024), 16, 0, 0);
    if (MODE != 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {  // 3 KiB of reads per KiB of DMA, like the f16 kernel (192 vs 64 KiB)
        const f4 v = *(const f4 *)(smem + 65536 + ((piece * 3 + r) & 63) * 1024 + lane * 16);
        acc += v;
      }
    }
    if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int MODE>
void run(const char *name, const char *src, float *out) {
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512, 131072>>>(src, out);
  hipEventRecord(e0);
  k<MODE><<<256, 512, 131072>>>(src, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double dma_bytes_per_cu = MODE == 1 ? 0 : (double)ITERS * 8 * 1024, rd = MODE == 0 ? 0 : (double)ITERS * 8 * 3 * 1024;
  const double clk = ms * 1e-3 * 2.4e9;
  printf("%-22s %8.3f ms   DMA %6.1f B/clk/CU   ds_read %6.1f B/clk/CU (2.4 GHz nominal)\n", name, ms, dma_bytes_per_cu / clk,
         rd / clk);
}
int main() {
  char *src; float *out;
  hipMalloc(&src, 256 * 65536); hipMemset(src, 1, 256 * 65536);
  hipMalloc(&out, 256 * 512 * 4);
  run<0>("DMA only", src, out); run<1>("ds_read_b128 only", src, out); run<2>("DMA + ds_read", src, out);
  return 0;
}
