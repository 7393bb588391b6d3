// Hardware probe: what does ds_read_b64_tr_b16 return?  LDS holds element index i at element i
// (16-bit).  Every lane passes the byte address 8*lane (experiment A) or the
// "row-major [4][16] block per 16-lane group" addresses (experiment B) and we print what each
// lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((vector_size(8)));
__global__ void probe(short *out, int mode) {
  __shared__ short lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int elem;
  if (mode == 0) elem = 4 * l;                                   // A: consecutive 8-B pieces
  else elem = (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4;  // B: [4][16] block per group, row stride 16
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lds + elem));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
