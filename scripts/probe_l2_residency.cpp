// Measurement probe (not product code): what does one launch of the recurrent step kernel pay to read (a) its read-only
// operand (packed W_hh: 128 slices x 96 KB, the C3 geometry) and (b) FRESH data written by the previous launch from all XCDs
// (the packed h_t / dGh operand)?  256 workgroups are dispatched from an own AQL queue with barrier-bit packets whose acquire
// scope is AGENT (what HIP uses) or NONE, so that the cost of the launch-boundary L2 invalidate can be separated.
//   usage: probe_l2_residency <hsaco> <slice KB> <exchange 0|1> <fresh KB> <mode>      (modes: see probe_l2_kernel.hip)
// Findings on MI355X (profiles/r01_probe_l2_residency.txt): the read-only set STAYS L2-resident across launches (acquire
// AGENT costs 0.15 us more than NONE); one fresh cross-XCD round trip costs ~0.9 us, 64 KB per workgroup ~1.6 us on top of the
// read-only slice when every load is in flight at once.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(s_, &m); printf("HSA error %s line %d\n", m, __LINE__); exit(1); } } while (0)

static hsa_status_t find_gpu(hsa_agent_t agent, void* data) {
  hsa_device_type_t t;
  hsa_agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && ((hsa_agent_t*)data)->handle == 0) *(hsa_agent_t*)data = agent;
  return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
  const char* hsaco = argc > 1 ? argv[1] : "scripts/build/probe_l2_kernel.hsaco";
  const int launches = 500;
  CK(hipSetDevice(0));
  CK(hipFree(0));
  HK(hsa_init());
  hsa_agent_t gpu = {0};
  HK(hsa_iterate_agents(find_gpu, &gpu));
  // ---- code object
  FILE* f = fopen(hsaco, "rb");
  if (!f) { printf("cannot open %s\n", hsaco); return 1; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> blob(sz);
  if (fread(blob.data(), 1, sz, f) != (size_t)sz) return 1;
  fclose(f);
  hsa_code_object_reader_t reader;
  HK(hsa_code_object_reader_create_from_memory(blob.data(), sz, &reader));
  hsa_executable_t exe;
  HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HK(hsa_executable_load_agent_code_object(exe, gpu, reader, nullptr, nullptr));
  HK(hsa_executable_freeze(exe, nullptr));
  const bool exchange = argc > 3 && atoi(argv[3]) != 0;
  hsa_executable_symbol_t sym;
  HK(hsa_executable_get_symbol_by_name(exe, exchange ? "read_w_exchange.kd" : "read_w.kd", &gpu, &sym));
  uint64_t kobj; uint32_t karg, lds, priv;
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &lds));
  HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
  // ---- data: 128 slices x 98304 B = 12.6 MB
  const int nslices = 128, slice_bytes = (argc > 2 ? atoi(argv[2]) : 96) * 1024, vec_per_wg = slice_bytes / 16;
  char* w; unsigned* sink; char* kargs;
  CK(hipMalloc(&w, (size_t)nslices * slice_bytes));
  CK(hipMemset(w, 0x11, (size_t)nslices * slice_bytes));
  CK(hipMalloc(&sink, 4096));
  char* xbuf;
  CK(hipMalloc(&xbuf, 2 * 4 * 65536));
  CK(hipMemset(xbuf, 0, 2 * 4 * 65536));
  const int fresh_kb = argc > 4 ? atoi(argv[4]) : 64;
  struct KA { const void* w; int vec_per_wg; int nslices; unsigned* sink; void* xbuf; int step; int fresh_vec; };
  CK(hipMalloc(&kargs, 512 * 500));                  // one 512-byte kernarg block per launch (>= the kernel's segment incl. hidden args)
  CK(hipMemset(kargs, 0, 512 * 500));
  for (int i = 0; i < 500; ++i) {
    KA ka = {w, vec_per_wg, nslices, sink, xbuf, i, fresh_kb * 64 | ((argc > 5 ? atoi(argv[5]) : 0) << 20)};
    CK(hipMemcpy(kargs + 512 * i, &ka, sizeof(ka), hipMemcpyHostToDevice));
  }
  printf("kernel object %llx kernarg %u B lds %u B scratch %u B\n", (unsigned long long)kobj, karg, lds, priv);
  CK(hipDeviceSynchronize());
  // ---- queue + signal
  hsa_queue_t* q;
  HK(hsa_queue_create(gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  hsa_signal_t done;
  HK(hsa_signal_create(1, 0, nullptr, &done));
  auto run = [&](int acquire_scope) {
    hsa_signal_store_relaxed(done, 1);
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t last = 0;
    for (int i = 0; i < launches; ++i) {
      const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
      hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
      p->workgroup_size_x = 512; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
      p->grid_size_x = 256 * 512; p->grid_size_y = 1; p->grid_size_z = 1;
      p->private_segment_size = priv; p->group_segment_size = lds;
      p->kernel_object = kobj; p->kernarg_address = kargs + 512 * i;
      p->completion_signal.handle = (i == launches - 1) ? done.handle : 0;
      const int acq = (i == 0) ? HSA_FENCE_SCOPE_AGENT : acquire_scope;
      const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                              (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
      const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
      __atomic_store_n((uint32_t*)p, header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
      last = idx;
    }
    hsa_signal_store_screlease(q->doorbell_signal, last);
    while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) {}
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return us / launches;
  };
  run(HSA_FENCE_SCOPE_AGENT);
  for (int rep = 0; rep < 3; ++rep) {
    const double a = run(HSA_FENCE_SCOPE_AGENT), n = run(HSA_FENCE_SCOPE_NONE);
    if (exchange) printf("[+ %d KB of data written by the previous launch, mode %d] ", fresh_kb, argc > 5 ? atoi(argv[5]) : 0);
    printf("rep %d: acquire AGENT %.2f us/launch | acquire NONE %.2f us/launch   (256 WGs x %d KB, %.1f MB working set, barrier-bit packets)\n", rep, a, n, slice_bytes / 1024,
           nslices * (double)slice_bytes / 1e6);
  }
  return 0;
}
