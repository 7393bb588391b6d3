// Hardware probe: which XCD a workgroup lands on, as a function of its linear id, for the two launch shapes the GEMM
// kernels use (512 threads / all of a CU's LDS = one workgroup per CU; 256 threads / 48 KiB = two per CU).  The tile
// rasterisation (mm_common.h: xcd_remap) assumes workgroup b -> XCD b % 8, "observed, used for speed only"; this prints
// how often that holds, and the order in which workgroups of one XCD start (s_memtime), on THIS box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void census(unsigned *xcc, unsigned long long *t0, int spin) {
  extern __shared__ char smem[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { xcc[blockIdx.x] = id & 0xf; t0[blockIdx.x] = t; }
  smem[threadIdx.x] = (char)id;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}

int run(const char *what, int blocks, int threads, int lds) {
  unsigned *d; unsigned long long *dt;
  CHECK(hipMalloc(&d, blocks * 4)); CHECK(hipMalloc(&dt, blocks * 8));
  CHECK(hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(census, dim3(blocks), dim3(threads), lds, 0, d, dt, 200);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned> h(blocks); std::vector<unsigned long long> ht(blocks);
  CHECK(hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(ht.data(), dt, blocks * 8, hipMemcpyDeviceToHost));
  int match = 0, hist[16] = {};
  for (int b = 0; b < blocks; ++b) { match += h[b] == (unsigned)(b % 8); hist[h[b]]++; }
  printf("%s: %d workgroups of %d threads, %d KiB LDS: xcc == b %% 8 for %.1f %% of them; per-XCC counts:", what, blocks, threads,
         lds >> 10, 100.0 * match / blocks);
  for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
  printf("\n  first 32 ids:");
  for (int b = 0; b < 32; ++b) printf(" %u", h[b]);
  // start order inside XCC 0: are its workgroups started in increasing b?
  std::vector<std::pair<unsigned long long, int>> v;
  for (int b = 0; b < blocks; ++b) if (h[b] == h[0]) v.push_back({ht[b], b});
  std::sort(v.begin(), v.end());
  int inversions = 0;
  for (size_t i = 1; i < v.size(); ++i) inversions += v[i].second < v[i - 1].second;
  printf("\n  XCC %u: %zu workgroups, %d start-order inversions; first started:", h[0], v.size(), inversions);
  for (size_t i = 0; i < 12 && i < v.size(); ++i) printf(" %d", v[i].second);
  printf("\n");
  hipFree(d); hipFree(dt);
  return 0;
}

int main() {
  run("one per CU ", 4096, 512, 160 * 1024);
  run("two per CU ", 8192, 256, 48 * 1024);
  run("two per CU ", 8192, 256, 64 * 1024);
  return 0;
}
