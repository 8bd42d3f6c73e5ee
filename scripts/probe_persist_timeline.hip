// Measurement probe: where does one time step of the PERSISTENT recurrence kernels (C3: GRU, H = 1024, B = 64, bf16 training mode) go?
// Builds the product's own rnn.hip with -DDS2_RNN_TRACE: every wave of workgroup 0 sums the s_memtime spans between consecutive stamps in
// registers (no memory traffic inside the time loop) and writes them out when the launch ends.
//   span 0: previous step's publish -> loop top        1: gather (tag polling)        2: HBM section + MFMAs + partial sums to LDS
//        3: workgroup barrier                          4: LDS sums + gate math        5: stage + publish + reset issued
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

int main(int argc, char** argv) {
  const int G = 3, H = argc > 1 ? atoi(argv[1]) : 1024, B = argc > 2 ? atoi(argv[2]) : 64, T = 501, bf = 1;
  const size_t M = (size_t)T * B;
  float *gx, *whh, *bhh, *hbuf, *aux, *dy;
  int* lens;
  void *wpf, *wpb, *ws, *wsb_, *rec, *dgx;
  CK(hipMalloc(&gx, M * 2 * G * H * 4)); CK(hipMalloc(&whh, (size_t)2 * G * H * H * 4)); CK(hipMalloc(&bhh, 2 * G * H * 4));
  CK(hipMalloc(&hbuf, M * 2 * H * 4)); CK(hipMalloc(&aux, M * 2 * H * 4)); CK(hipMalloc(&lens, B * 4)); CK(hipMalloc(&dy, M * H * 4));
  CK(hipMalloc(&rec, M * 2 * H * 8)); CK(hipMalloc(&dgx, M * 2 * G * H * 2));
  std::vector<float> h((size_t)2 * G * H * H);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((int)(i * 2654435761u >> 8) % 2001 - 1000) * (1.0f / 32000.0f);
  CK(hipMemcpy(whh, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bhh, 0, 2 * G * H * 4));
  CK(hipMemset(gx, 0, M * 2 * G * H * 4));
  CK(hipMemset(dy, 0, M * H * 4));
  std::vector<int> hl(B, T);
  CK(hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&wpf, ds2_rnn_packed_bytes(G, H, 0, bf))); CK(hipMalloc(&wpb, ds2_rnn_packed_bytes(G, H, 1, bf)));
  if (ds2_rnn_pack_whh(G, whh, wpf, wpb, H, bf, nullptr)) { printf("pack failed: %s\n", ds2_last_error()); return 1; }
  const size_t wsf = ds2_rnn_fwd_workspace_bytes(B, H, bf), wsb = ds2_rnn_bwd_workspace_bytes(G, B, H, bf);
  CK(hipMalloc(&ws, wsf)); CK(hipMalloc(&wsb_, wsb));
  unsigned long long* trace;
  CK(hipMalloc(&trace, 2 * NW * 8 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rnn_trace), &trace, sizeof(trace)));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float msf = 0, msb = 0;
  ds2_rnn_ctx ctx;                                       // caller-owned recurrence context (round 5's C ABI)
  int* status_dev;
  CK(hipMalloc(&status_dev, 8 * sizeof(int))); CK(hipMemset(status_dev, 0, 8 * sizeof(int)));
  if (ds2_rnn_ctx_init(&ctx, status_dev, nullptr, nullptr)) { printf("ctx init failed: %s\n", ds2_last_error()); return 1; }
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(trace, 0, 2 * NW * 8 * 8));
    CK(hipEventRecord(e0));
    if (ds2_rnn_fwd(&ctx, G, gx, wpf, bhh, hbuf, aux, lens, T, B, H, bf, rec, ws, wsf, nullptr)) { printf("fwd failed: %s\n", ds2_last_error()); return 1; }
    CK(hipEventRecord(e1));
    if (ds2_rnn_bwd(&ctx, G, dy, H, nullptr, aux, hbuf, wpb, lens, T, B, H, bf, dgx, rec, wsb_, wsb, nullptr)) { printf("bwd failed: %s\n", ds2_last_error()); return 1; }
    CK(hipEventRecord(e2));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&msf, e0, e1)); CK(hipEventElapsedTime(&msb, e1, e2));
  }
  int st[8];
  ds2_rnn_persistent_status(&ctx, st);
  printf("paths taken: %d (3 = both persistent, 7 = K-split backward), starved %d\n", ds2_rnn_last_path(&ctx), st[0]);
  std::vector<unsigned long long> tr(2 * NW * 8);
  CK(hipMemcpy(tr.data(), trace, tr.size() * 8, hipMemcpyDeviceToHost));
  const char* names_ag[7] = {"publish -> loop top", "gather (poll)", "HBM issue + MFMA + partials", "barrier", "LDS sums + gate math", "stage + publish", "-"};
  // K-split backward kernel (rnn_bwd_ksplit.h; the default where the shape qualifies, DS2_RNN_KSPLIT=0 selects the all-gather kernel)
  const char* names_ks[7] = {"store issue -> loop top", "gather (poll) + reduce-scatter", "next fetch issued + gate math", "dGh -> LDS + barrier", "LDS read + MFMA issue",
                             "MFMA drain + pack + publish", "result stores + offset step"};
  for (int kind = 0; kind < 2; ++kind) {
    const char** names = (kind == 1 && (ds2_rnn_last_path(&ctx) & 4)) ? names_ks : names_ag;
    const double us_step = (kind ? msb : msf) * 1e3 / T;
    double tot = 0;
    for (int k = 0; k < 7; ++k) tot += (double)tr[(kind * NW + 0) * 8 + k];
    const double tick = us_step * T / tot;     // ticks -> us, from wave 0's total
    printf("%s: %.2f us/step (events, traced build); wave 0 total %.0f ticks -> %.4f ns/tick\n", kind ? "BACKWARD" : "FORWARD", us_step, tot, tick * 1e3);
    for (int k = 0; k < 7; ++k) {
      printf("  %-32s", names[k]);
      for (int w = 0; w < NW; ++w) printf(" %5.2f", (double)tr[(kind * NW + w) * 8 + k] / T * tick);
      printf("  us (waves 0..7)\n");
    }
    printf("  %-32s", "poll passes per step");
    for (int w = 0; w < NW; ++w) printf(" %5.2f", (double)tr[(kind * NW + w) * 8 + 7] / (T - 1));
    printf("  (waves 0..7)\n");
  }
  return 0;
}
