// Where do conv1's bf16 kernels spend their time?  The product source compiled with -DDS2_C1_TRACE: wave 0 of every block sums the shader
// clocks of the phases of its main loop.   build: scripts/build_probes.sh;  run: scripts/build/probe_conv1
#define DS2_C1_TRACE
#include "../asr_amd/csrc/conv1_bf16.hip"
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int B = 64, F = 161, Tin = 1001, T = 501, D1 = 81;
  std::vector<float> hx((size_t)B * F * Tin);
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : hx) v = nd(rng);
  float *x, *w, *bias, *y, *dw; void *X16, *X16T, *wp, *ws; int* lens;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&X16, ds2_conv1_bf16_bytes(1, B, F, T))); CK(hipMalloc(&X16T, ds2_conv1_bf16_bytes(2, B, F, T))); CK(hipMalloc(&wp, ds2_conv1_bf16_bytes(0, B, F, T)));
  CK(hipMalloc(&w, 32 * 41 * 11 * 4)); CK(hipMemcpy(w, hx.data(), 32 * 41 * 11 * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&bias, 128)); CK(hipMemset(bias, 0, 128)); CK(hipMalloc(&dw, 32 * 41 * 11 * 4));
  CK(hipMalloc(&y, (size_t)B * 32 * D1 * T * 4));
  std::vector<int> hl(B, T); CK(hipMalloc(&lens, B * 4)); CK(hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice));
  const size_t wsb = ds2_conv1_wgrad_bf16_workspace_bytes(B, Tin); CK(hipMalloc(&ws, wsb));
  if (ds2_conv1_pack_bf16(w, wp, nullptr) || ds2_conv1_gather_bf16(x, X16, X16T, B, F, Tin, nullptr)) { printf("setup failed\n"); return 1; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      int rc = which == 0 ? ds2_conv1_fwd_bf16(X16, wp, bias, lens, y, B, F, Tin, nullptr) : ds2_conv1_wgrad_bf16(X16T, y, lens, dw, B, F, Tin, ws, wsb, nullptr);
      if (rc) { printf("launch failed\n"); return 1; }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> tr(8192 * 8);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_c1_trace), tr.size() * 8));
    const int base = which == 0 ? 0 : 4096, nblk = which == 0 ? ds2_conv1_fwd_bf16_stat_blocks(B, F, Tin) : 8 * B;
    double sum[8] = {0};
    for (int b = 0; b < nblk; ++b) for (int k = 0; k < 8; ++k) sum[k] += (double)tr[(size_t)(base + b) * 8 + k];
    printf("%s: %.1f us (with the trace's s_memtime reads), %d blocks; mean shader clocks per block and phase:", which == 0 ? "conv1 fwd" : "conv1 wgrad", ms * 1e3, nblk);
    for (int k = 0; k < 6; ++k) printf("  [%d] %.0f", k, sum[k] / nblk);
    printf("\n");
  }
  printf("fwd phases: 0 prologue, 1 MFMAs + fragment reads, 2 vmcnt(0), 3 barrier, 4 DMA issue, 5 stores; 8 tiles per block\n");
  printf("wgrad phases: 0 prologue, 1 fetch (9 loads), 2 compute, 3 publish, 4 barrier; 81 steps per block\n");
  return 0;
}
