// Probe: lane semantics of ds_read_b64_tr_b16 on gfx950.
// LDS holds 16-bit element e at index e; lane l reads the 8-byte datum at
// byte address 8*perm(l).  Prints the four elements every lane received.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_v4;
__global__ void k(short* out) {
  __shared__ short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 512);
  k<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l)
    printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  // expected under the assumed rule: lane i, reg j <- element (i&3) of the datum of lane (16*(i>>4) + 4j + ((i&15)>>2))
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    int src = 16 * (l >> 4) + 4 * j + ((l & 15) >> 2);
    int want = 4 * src + (l & 3);
    if (h[4*l+j] != want) ++bad;
  }
  printf("assumed rule mismatches: %d\n", bad);
  return 0;
}
