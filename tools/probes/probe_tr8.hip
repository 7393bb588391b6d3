// Hardware probe: ds_read_b64_tr_b8 semantics.  LDS byte i holds (i & 0xff); lane l passes byte
// address 8*l; prints the 8 bytes each lane receives.  Also checks the i8 MFMA operand layout with
// a one-hot experiment: A = e(row r0, k k0), B = e(k k0, col c0) -> D[r0][c0] = 1 tells which lane
// /byte supplies which (row/col, k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int v2i __attribute__((vector_size(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void probe_tr(unsigned char *out) {
  __shared__ unsigned char lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned char)(i & 0xff);
  __syncthreads();
  v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lds + threadIdx.x * 8));
  unsigned char b[8];
  memcpy(b, &v, 8);
  for (int j = 0; j < 8; ++j) out[threadIdx.x * 8 + j] = b[j];
}
// lane la supplies byte ba of A = 1, lane lb supplies byte bb of B = 1; report which D element is 1
__global__ void probe_mfma(int *out, int la, int ba, int lb, int bb) {
  v4i a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if ((int)threadIdx.x == la) a[ba / 4] = 1 << (8 * (ba % 4));
  if ((int)threadIdx.x == lb) b[bb / 4] = 1 << (8 * (bb % 4));
  v16i c = {};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = c[r];
}
int main() {
  unsigned char *d, h[512];
  hipMalloc(&d, 512);
  probe_tr<<<1, 64>>>(d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 20; ++l) {
    printf("tr8 lane %2d:", l);
    for (int j = 0; j < 8; ++j) printf(" %3d", h[l * 8 + j]);
    printf("\n");
  }
  int *di, hi[1024];
  hipMalloc(&di, sizeof(hi));
  // k index of (lane, byte): find pairs that hit: A lane la byte ba multiplies with B lane lb byte bb iff same k
  const int tests[][4] = {{0, 0, 0, 0}, {0, 5, 0, 5}, {0, 5, 0, 6}, {0, 15, 0, 15}, {32, 0, 32, 0}, {0, 0, 32, 0},
                          {3, 7, 9, 7}, {35, 9, 41, 9}, {3, 7, 41, 7}};
  for (auto &t : tests) {
    probe_mfma<<<1, 64>>>(di, t[0], t[1], t[2], t[3]);
    hipMemcpy(hi, di, sizeof(hi), hipMemcpyDeviceToHost);
    int hits = 0, hl = -1, hr = -1;
    for (int i = 0; i < 1024; ++i) if (hi[i]) { ++hits; hl = i / 16; hr = i % 16; }
    printf("A(lane %2d byte %2d) x B(lane %2d byte %2d): hits %d at lane %d reg %d  -> row %d col %d\n", t[0], t[1], t[2], t[3],
           hits, hl, hr, hl < 0 ? -1 : (hr & 3) + 8 * (hr >> 2) + 4 * (hl >> 5), hl < 0 ? -1 : hl & 31);
  }
  return 0;
}
