// The in-wave reduce-scatter of the K-split backward recurrence (asr_amd/csrc/rnn_bwd_ksplit.h): v_permlane32_swap / v_permlane16_swap / DPP
// row_ror:8 against the same tree written with ds_bpermute shuffles, and against the plain sum over the 8 lanes l ^ {8,16,32 combos}.
#include <hip/hip_runtime.h>
#include "../asr_amd/csrc/permlane.h"
#include <cstdio>
__global__ void k(const float* in, float* out_fast, float* out_ref) {
  const int lane = threadIdx.x & 63, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = lane >> 5;
  float S[8];
  for (int v = 0; v < 8; ++v) S[v] = in[lane * 8 + v];
  float R[4], Q[2], fast, ref;
  for (int kk = 0; kk < 4; ++kk) {
    const u32pair r = permlane32_swap(S[kk], S[kk + 4]);
    R[kk] = r.a + r.b;
  }
  for (int kk = 0; kk < 2; ++kk) {
    const u32pair r = permlane16_swap(R[kk], R[kk + 2]);
    Q[kk] = r.a + r.b;
  }
  {
    const float keep = b3 ? Q[1] : Q[0], send = b3 ? Q[0] : Q[1];
    const int got = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xf, 0xf, false);
    fast = keep + __builtin_bit_cast(float, got);
  }
  // reference: value v = 4 b5 + 2 b4 + b3 summed over the 8 lanes that share lane & 7
  const int v = 4 * b5 + 2 * b4 + b3;
  ref = 0.f;
  for (int o = 0; o < 8; ++o) ref += __shfl(S[0] * 0.f + in[((lane & 7) + 8 * o) * 8 + v], lane, 64);
  out_fast[lane] = fast;
  out_ref[lane] = ref;
  // stage outputs for diagnosis
  out_fast[64 + lane] = R[0]; out_fast[128 + lane] = Q[0];
}
int main() {
  float h[512], *d, *f, *r, hf[192], hr[64];
  for (int i = 0; i < 512; ++i) h[i] = (float)((i * 37) % 101) + 0.25f * (i % 7);   // exactly representable small values: sums are exact
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&f, sizeof(hf)); (void)hipMalloc(&r, sizeof(hr));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, f, r);
  (void)hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost); (void)hipMemcpy(hr, r, sizeof(hr), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) if (hf[l] != hr[l]) { if (bad < 8) printf("lane %d fast %g ref %g\n", l, hf[l], hr[l]); ++bad; }
  // what R[0] holds: expected lanes<32: S0[l] + S0[l+32]; lanes>=32: S4[l-32] + S4[l]
  int badR = 0;
  for (int l = 0; l < 64; ++l) {
    const float e = l < 32 ? h[l * 8 + 0] + h[(l + 32) * 8 + 0] : h[(l - 32) * 8 + 4] + h[l * 8 + 4];
    if (hf[64 + l] != e) { if (badR < 4) printf("R0 lane %d got %g expected %g\n", l, hf[64 + l], e); ++badR; }
  }
  printf("reduce-scatter: %d of 64 lanes differ from the reference sum; stage A: %d differ\n", bad, badR);
  return bad != 0;
}
