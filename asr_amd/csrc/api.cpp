// Error/version plumbing of the C-ABI (include/ds2hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_ds2_err[1024] = "";

int ds2_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_ds2_err, sizeof(g_ds2_err), fmt, ap);
  va_end(ap);
  return -1;
}

extern "C" const char* ds2_last_error(void) { return g_ds2_err; }
extern "C" const char* ds2_version(void) { return "ds2hip 0.1.0 (gfx950)"; }

extern "C" int ds2_device_info(int* cu_count, int* wave_size, char* arch, int arch_len) {
  hipDeviceProp_t p;
  int dev = 0;
  DS2_HIP(hipGetDevice(&dev));
  DS2_HIP(hipGetDeviceProperties(&p, dev));
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", p.gcnArchName);
  return 0;
}

// Kernel-family / tile-shape SELECTORS of the recurrence (every selection computes the full result): 8 / 16 = alternative tile shapes of the
// wide step kernels, 64 = step kernels instead of the persistent ones, 128 = all-gather persistent backward instead of the K-split one.
// Bits 1 / 2 (skip the h.W_hh product / the gate epilogue: scripts/ablate_rnn.py) SKIP WORK and exist only in a library built with
// -DDS2_ABLATE (make ABLATE=1); the shipped library masks them off here and compiles the tests out of the kernels (rnn.hip).
extern "C" int ds2_debug_flags(ds2_rnn_ctx* ctx, int flags) {
  if (!ctx) return ds2_set_error("ds2_debug_flags: null context");
  int old = ctx->debug_flags;
#ifndef DS2_ABLATE
  flags &= ~3;
#endif
  ctx->debug_flags = flags;
  return old;
}
// The recurrence's state lives with the caller (include/ds2hip.h): zero the struct, record where its device / pinned words are.
extern "C" int ds2_rnn_ctx_init(ds2_rnn_ctx* ctx, int* status_dev, int* poison_host, int* poison_dev) {
  if (!ctx || !status_dev) return ds2_set_error("ds2_rnn_ctx_init: ctx and status_dev are required");
  if ((poison_host == nullptr) != (poison_dev == nullptr)) return ds2_set_error("ds2_rnn_ctx_init: poison_host and poison_dev go together");
  memset(ctx, 0, sizeof(*ctx));
  ctx->size = (int)sizeof(*ctx);
  ctx->persist_fwd = ctx->persist_bwd = 1;
  const char* env = getenv("DS2_RNN_REARM_CALLS");            // read here, once per context; never written
  ctx->rearm_calls = env ? atoi(env) : 64;
  ctx->status_dev = status_dev; ctx->poison_host = poison_host; ctx->poison_dev = poison_dev;
  return 0;
}
extern "C" int ds2_memset_async(void* dst, int value, size_t bytes, void* stream) {
  if (!dst || bytes == 0) return ds2_set_error("ds2_memset_async: bad arguments");
  hipError_t e = hipMemsetAsync(dst, value, bytes, (hipStream_t)stream);
  return e == hipSuccess ? 0 : ds2_set_error("ds2_memset_async: %s", hipGetErrorString(e));
}
extern "C" int ds2_ablation_build(void) {
#ifdef DS2_ABLATE
  return 1;
#else
  return 0;
#endif
}
