// Greedy CTC decode on the device: per-frame arg-max, collapse repeats, drop blanks, with the frame
// offset of every kept label.  Replaces GreedyDecoder.decode -> convert_to_strings -> process_string
// (asr_deepspeech/decoders/greedy_decoder.py:10-68), which walks every frame on the host with one
// `.item()` sync each.  The host side only maps the compacted label ids to characters.
//
//   probs (B,T,C) fp32, element strides ld_b / ld_t, C contiguous (probabilities or logits: the
//         arg-max is the same); ties resolve to the LOWEST class index (torch.max's first-maximum rule)
//   sizes (B) int32 valid frames per utterance, or null = T
//   out   ids (B,T) int32 kept labels, offs (B,T) int32 their frame index, out_len (B) int32
//
// Kernel 1: one wavefront per (b,t) row, lanes stride the classes, (value,index) butterfly.
// Kernel 2: one wavefront per utterance: 64 frames at a time, keep-flag ballot + popcount prefix.
// HBM-bound byte work: B*T*C*4 bytes read once (7.4 MB for 64 x 1001 x 29), nothing re-read.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ probs, long long ld_b, long long ld_t, int Bn,
                                                          int T, int C, const int* __restrict__ sizes, int* __restrict__ frame_ids) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Bn * T) return;
  const int b = row / T, t = row - b * T;
  if (sizes && t >= sizes[b]) return;
  const int lane = threadIdx.x & 63;
  const float* p = probs + b * ld_b + t * ld_t;
  float best = -INFINITY;
  int arg = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = p[c];
    if (v > best || arg == 0x7fffffff) {  // strictly greater keeps the first maximum within a lane's stride
      best = v;
      arg = c;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oa = __shfl_xor(arg, off, 64);
    if (ov > best || (ov == best && oa < arg)) {
      best = ov;
      arg = oa;
    }
  }
  if (lane == 0) frame_ids[row] = arg;
}

__global__ __launch_bounds__(64) void collapse_kernel(const int* __restrict__ frame_ids, int T, const int* __restrict__ sizes,
                                                      int blank, int* __restrict__ ids, int* __restrict__ offs,
                                                      int* __restrict__ out_len) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = sizes ? min(max(sizes[b], 0), T) : T;
  const int* f = frame_ids + (long long)b * T;
  int count = 0;
  for (int base = 0; base < n; base += 64) {
    const int t = base + lane;
    int id = blank, prev = blank;
    if (t < n) {
      id = f[t];
      prev = t > 0 ? f[t - 1] : -1;
    }
    const bool keep = t < n && id != blank && (t == 0 || id != prev);
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
      ids[(long long)b * T + pos] = id;
      offs[(long long)b * T + pos] = t;
    }
    count += __popcll(m);
  }
  if (lane == 0) out_len[b] = count;
}

}  // namespace

extern "C" size_t ds2_greedy_decode_workspace_bytes(int B, int T) { return (size_t)B * T * sizeof(int); }

extern "C" int ds2_greedy_decode_f32(const float* probs, long long ld_b, long long ld_t, int B, int T, int C, const int* sizes_dev,
                                     int blank, int* ids, int* offs, int* out_len, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(probs && ids && offs && out_len, "ds2_greedy_decode_f32: null pointer");
  DS2_REQUIRE(B > 0 && T > 0 && C > 0 && blank >= 0 && blank < C, "ds2_greedy_decode_f32: bad dims (B=%d T=%d C=%d blank=%d)", B, T, C,
              blank);
  DS2_REQUIRE(ws && ws_bytes >= ds2_greedy_decode_workspace_bytes(B, T), "ds2_greedy_decode_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int* frame_ids = (int*)ws;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(ceil_div(B * T, 4)), dim3(256), 0, s, probs, ld_b, ld_t, B, T, C, sizes_dev, frame_ids);
  DS2_LAUNCH_CHECK("argmax_rows_kernel");
  hipLaunchKernelGGL(collapse_kernel, dim3(B), dim3(64), 0, s, (const int*)frame_ids, T, sizes_dev, blank, ids, offs, out_len);
  DS2_LAUNCH_CHECK("collapse_kernel");
  return 0;
}
