// Measurement probe: where does one forward recurrent step (C3: H = 1024, B = 64, bf16 operands) spend its ~6 us?
// Builds the product's own rnn.hip with -DDS2_RNN_TRACE (s_memtime stamps in workgroup (0,0,0), lane 0 of each wave):
//   0 kernel entry | 1 kernel arguments in SGPRs | 2 MFMAs done (all operand loads consumed) | 3 partials in LDS + barrier
//   4 epilogue done, stores issued | 5 stores acknowledged (vmcnt(0))
// and prints the mean span of each phase plus the entry-to-entry period of consecutive launches.
#include "../asr_amd/csrc/rnn.hip"
#include <cstdio>
// (rnn.hip's ds2_rnn_bwd_bn falls back to this norm.hip entry; the probe links rnn.hip alone and never calls it)
int ds2i_bn1d_bwd_apply(const float*, int, const float*, int, float*, int, int, int, const float*, const float*, const float*, const float*, const float*,
                        float, hipStream_t) { return -1; }
int ds2i_bn1d_bwd_apply_xbf(const float*, int, const void*, int, float*, int, int, int, const float*, const float*, const float*, const float*, const float*,
                            float, hipStream_t) { return -1; }
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
  const int G = 3, H = 1024, B = 64, T = 501, bf = 1;
  const size_t M = (size_t)T * B;
  float *gx, *whh, *bhh, *hbuf, *aux;
  int* lens;
  void *wpf, *wpb, *ws;
  CK(hipMalloc(&gx, M * 2 * G * H * 4)); CK(hipMalloc(&whh, (size_t)2 * G * H * H * 4)); CK(hipMalloc(&bhh, 2 * G * H * 4));
  CK(hipMalloc(&hbuf, M * 2 * H * 4)); CK(hipMalloc(&aux, M * 2 * H * 4)); CK(hipMalloc(&lens, B * 4));
  std::vector<float> h((size_t)2 * G * H * H);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((int)(i * 2654435761u >> 8) % 2001 - 1000) * (1.0f / 32000.0f);
  CK(hipMemcpy(whh, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bhh, 0, 2 * G * H * 4));
  CK(hipMemset(gx, 0, M * 2 * G * H * 4));
  std::vector<int> hl(B, T);
  CK(hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&wpf, ds2_rnn_packed_bytes(G, H, 0, bf))); CK(hipMalloc(&wpb, ds2_rnn_packed_bytes(G, H, 1, bf)));
  if (ds2_rnn_pack_whh(G, whh, wpf, wpb, H, bf, nullptr)) { printf("pack failed: %s\n", ds2_last_error()); return 1; }
  const size_t wsb = ds2_rnn_fwd_workspace_bytes(B, H, bf);
  CK(hipMalloc(&ws, wsb));
  unsigned long long* trace;
  CK(hipMalloc(&trace, (size_t)T * NW * 8 * 8));
  CK(hipMemset(trace, 0, (size_t)T * NW * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rnn_trace), &trace, sizeof(trace)));
  for (int rep = 0; rep < 2; ++rep)
    if (ds2_rnn_fwd(G, gx, wpf, bhh, hbuf, aux, lens, T, B, H, bf, nullptr, ws, wsb, nullptr)) { printf("fwd failed: %s\n", ds2_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> tr((size_t)T * NW * 8);
  CK(hipMemcpy(tr.data(), trace, tr.size() * 8, hipMemcpyDeviceToHost));
  // 100 MHz s_memtime? report in raw ticks and, using the period measured by events, in us
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  ds2_rnn_fwd(G, gx, wpf, bhh, hbuf, aux, lens, T, B, H, bf, nullptr, ws, wsb, nullptr);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us_per_step = ms * 1e3 / T;
  double span[8] = {0, 0, 0, 0, 0, 0, 0, 0}, period = 0;
  int n = 0;
  for (int s = 50; s < T - 1; ++s) {
    unsigned long long mn[8], mx[8];
    for (int k = 0; k < 8; ++k) { mn[k] = ~0ull; mx[k] = 0; }
    for (int w = 0; w < NW; ++w)
      for (int k = 0; k < 8; ++k) {
        const unsigned long long v = tr[((size_t)s * NW + w) * 8 + k];
        if (v < mn[k]) mn[k] = v;
        if (v > mx[k]) mx[k] = v;
      }
    unsigned long long next0 = ~0ull;
    for (int w = 0; w < NW; ++w) { const unsigned long long v = tr[((size_t)(s + 1) * NW + w) * 8 + 0]; if (v < next0) next0 = v; }
    // persistent kernel stamps: 0 step start, 1 gather done (operand polled in), 2 MFMAs done + partial sums written, 3 barrier A passed,
    // 6 LDS sums + gate math done, 7 barrier B passed (piece staged), 4 end of the step's stores (next step starts)
    span[0] += (double)(mx[1] - mn[0]);
    span[1] += (double)(mx[2] - mx[1]);
    span[2] += (double)(mx[3] - mx[2]);
    span[3] += (double)(mx[6] - mx[3]);
    span[4] += (double)(mx[7] - mx[6]);
    span[5] += (double)(mx[4] - mx[7]);
    span[6] += 0;
    span[7] += (double)(next0 - mx[4]);
    period += (double)(next0 - mn[0]);
    ++n;
  }
  const double tick_us = us_per_step / (period / n);
  printf("period per step: %.2f us (HIP events, traced build) = %.1f ticks -> 1 tick = %.4f us\n", us_per_step, period / n, tick_us);
  const char* names[8] = {"step start -> operand gathered (tag polling)", "-> MFMAs done, partial sums written", "-> barrier A passed",
                          "-> LDS sums + gate math done", "-> piece staged, barrier B passed", "-> publish + HBM stores issued",
                          "(unused)", "-> next step starts"};
  for (int k = 0; k < 8; ++k) printf("  %-56s %6.2f us\n", names[k], span[k] / n * tick_us);
  return 0;
}
