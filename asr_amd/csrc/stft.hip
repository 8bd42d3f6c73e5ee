// Spectrogram front-end on the GPU: the step before the train path.
// Replaces SpectrogramParser.parse_audio's arithmetic (asr_deepspeech/data/parsers/spectrogram_parser.py:45-60):
//     D = librosa.stft(y, n_fft, hop_length, win_length=n_fft, window)      (centred frames)
//     spect = log1p(|D|) ; optional (spect - mean) / std (unbiased, over the whole utterance)
// for a whole batch of waveforms at once, writing the (B,1,n_bins,T) zero-padded layout _collate_fn builds
// (functional.py:18-30).  librosa (0.11.0 in the reference's uv.lock) is a third-party dependency that is not in the
// reference tree; its published algorithm is restated: pad n_fft/2 on both sides (zeros = librosa >= 0.10 default, or
// reflect), frame t = padded[t*hop : t*hop + n_fft] * window, one-sided DFT, frames = 1 + n_samples / hop.
//
// Mapping: n_fft = 320 is not a power of two and tiny, so the DFT is a GEMM on the f32 matrix cores: all frames of all
// utterances are the rows of ONE (B*R, n_fft) operand with row pitch = hop — overlapping rows, no frame copy — times the
// window-folded basis (n_fft, 2*n_bins).  Around it: one padding pass, one LDS-tiled magnitude/log1p/transpose pass that
// also produces the per-utterance sums, and one normalisation pass.  HBM: ~2 x 82 MB at B=64 x 10 s; 13 GFLOP.
#include "common.h"

namespace {

// ypad[b*R*hop + i] = padded waveform of utterance b (i in [0, R*hop)), zero beyond n_b + 2*half
__global__ __launch_bounds__(256) void stft_pad_kernel(const float* __restrict__ audio, long long ld_audio, const int* __restrict__ n_samples,
                                                       float* __restrict__ ypad, int Bn, long long row_len, int half, int reflect, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    const long long b = i / row_len;
    if (b < Bn) {
      const int n = n_samples[b];
      long long j = i - b * row_len - half;            // index into the un-padded waveform
      if (j >= -(long long)half && j < (long long)n + half && n > 0) {
        if (j < 0) j = reflect ? -j : -1;
        else if (j >= n) j = reflect ? 2LL * (n - 1) - j : -1;
        if (j >= 0 && j < n) v = audio[b * ld_audio + j];
      }
    }
    ypad[i] = v;
  }
}

// C (B*R, 2*nb) [re, im interleaved per bin] -> out (B, nb, T) = log1p(sqrt(re^2 + im^2)) for t < frames_b else 0 ;
// part[b][blk][2] = (sum, sum of squares) over the valid elements of this block's tile.   block = 64 frames x 32 bins.
__global__ __launch_bounds__(256) void stft_post_kernel(const float* __restrict__ C, int ldc, int R, int nb, int T, int hop,
                                                        const int* __restrict__ n_samples, float* __restrict__ out, float* __restrict__ part) {
  __shared__ float tile[64][33];
  __shared__ float red[2][4];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, k0 = blockIdx.y * 32;
  const int n = n_samples[b];
  const int frames = n > 0 ? min(T, 1 + n / hop) : 0;
  const int tid = threadIdx.x;
  float s = 0.f, s2 = 0.f;
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int tl = pass * 8 + (tid >> 5), kl = tid & 31;
    const int t = t0 + tl, k = k0 + kl;
    float v = 0.f;
    if (t < frames && k < nb) {
      const float2 z = *reinterpret_cast<const float2*>(C + ((long long)b * R + t) * ldc + 2 * k);
      v = log1pf(sqrtf(z.x * z.x + z.y * z.y));
      s += v;
      s2 += v * v;
    }
    tile[tl][kl] = v;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int kl = pass * 4 + (tid >> 6), tl = tid & 63;
    const int t = t0 + tl, k = k0 + kl;
    if (t < T && k < nb) out[((long long)b * nb + k) * T + t] = tile[tl][kl];
  }
  s = wave_sum(s);
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = s2; }
  __syncthreads();
  if (tid == 0) {
    const long long blk = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    float* p = part + ((long long)b * gridDim.x * gridDim.y + blk) * 2;
    p[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    p[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// per utterance: mean and 1/std (unbiased) from the block partials, combined in fp64
__global__ void stft_stats_kernel(const float* __restrict__ part, int nblk, int nb, int T, int hop, const int* __restrict__ n_samples,
                                  float* __restrict__ stats) {
  const int b = blockIdx.x;
  __shared__ double sh[2][64];
  double s = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) {
    s += (double)part[((long long)b * nblk + i) * 2];
    s2 += (double)part[((long long)b * nblk + i) * 2 + 1];
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = 0.0; s2 = 0.0;
    for (int i = 0; i < 64; ++i) { s += sh[0][i]; s2 += sh[1][i]; }
    const int n = n_samples[b];
    const double cnt = n > 0 ? (double)nb * (double)min(T, 1 + n / hop) : 0.0;
    const double mean = cnt > 0 ? s / cnt : 0.0;
    const double var = cnt > 1 ? fmax(s2 - cnt * mean * mean, 0.0) / (cnt - 1.0) : 0.0;
    stats[2 * b] = (float)mean;
    stats[2 * b + 1] = var > 0 ? (float)(1.0 / sqrt(var)) : 0.f;
  }
}

// out[b][k][t] = (out - mean_b) * rstd_b for t < frames_b (padding stays 0: _collate_fn pads AFTER normalisation)
__global__ __launch_bounds__(256) void stft_normalize_kernel(float* __restrict__ out, int nb, int T, int hop, const int* __restrict__ n_samples,
                                                             const float* __restrict__ stats, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int b = (int)(i / ((long long)nb * T));
    const int n = n_samples[b];
    const int frames = n > 0 ? min(T, 1 + n / hop) : 0;
    if (t < frames) out[i] = (out[i] - stats[2 * b]) * stats[2 * b + 1];
  }
}

inline int rows_per_utt(int T, int n_fft, int hop) { return T + ceil_div(n_fft, hop) - 1; }

}  // namespace

// frames an utterance of n samples produces (librosa centred STFT): 1 + n / hop
extern "C" int ds2_spectrogram_frames(int n_samples, int hop) { return n_samples > 0 ? 1 + n_samples / hop : 0; }

extern "C" size_t ds2_spectrogram_workspace_bytes(int B, int T, int n_fft, int hop) {
  const size_t R = (size_t)rows_per_utt(T, n_fft, hop);
  const size_t nb = (size_t)n_fft / 2 + 1;
  const size_t ypad = align_up(((size_t)B * R * hop + n_fft) * sizeof(float), 256);
  const size_t c = align_up((size_t)B * R * 2 * nb * sizeof(float), 256);
  const size_t part = align_up((size_t)B * ceil_div(T, 64) * ceil_div((int)nb, 32) * 2 * sizeof(float), 256);
  return ypad + c + part + align_up((size_t)B * 2 * sizeof(float), 256);
}

//   audio      (B, ld_audio) fp32 waveforms on the device, n_samples_dev (B) int32 valid samples per row
//   basis      (n_fft, 2*n_bins) fp32, basis[k][2j] = w[k] cos(2 pi k j / n_fft), basis[k][2j+1] = -w[k] sin(2 pi k j / n_fft)
//   out        (B, n_bins, T) fp32 (= (B,1,n_bins,T)); T >= max frames; frames beyond each utterance's own are 0
//   pad_mode   0 = zeros (librosa >= 0.10 default), 1 = reflect ; normalize 0 | 1
extern "C" int ds2_spectrogram_f32(const float* audio, long long ld_audio, const int* n_samples_dev, int B, int T, int n_fft, int hop,
                                   const float* basis, int pad_mode, int normalize, float* out, void* ws, size_t ws_bytes, void* stream) {
  DS2_REQUIRE(audio && n_samples_dev && basis && out && ws, "ds2_spectrogram_f32: null pointer");
  DS2_REQUIRE(B > 0 && T > 0 && n_fft >= 4 && (n_fft % 2) == 0 && hop > 0 && (hop % 4) == 0 && hop <= n_fft,
              "ds2_spectrogram_f32: bad dims (B=%d T=%d n_fft=%d hop=%d; hop must be a multiple of 4)", B, T, n_fft, hop);
  DS2_REQUIRE(ws_bytes >= ds2_spectrogram_workspace_bytes(B, T, n_fft, hop), "ds2_spectrogram_f32: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int R = rows_per_utt(T, n_fft, hop), nb = n_fft / 2 + 1;
  const long long row_len = (long long)R * hop;
  const long long ypad_n = (long long)B * row_len + n_fft;
  char* w = (char*)ws;
  float* ypad = (float*)w;                 w += align_up((size_t)ypad_n * sizeof(float), 256);
  float* C = (float*)w;                    w += align_up((size_t)B * R * 2 * nb * sizeof(float), 256);
  float* part = (float*)w;                 w += align_up((size_t)B * ceil_div(T, 64) * ceil_div(nb, 32) * 2 * sizeof(float), 256);
  float* stats = (float*)w;
  int blocks = (int)((ypad_n + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(stft_pad_kernel, dim3(blocks), dim3(256), 0, s, audio, ld_audio, n_samples_dev, ypad, B, row_len, n_fft / 2, pad_mode, ypad_n);
  DS2_LAUNCH_CHECK("stft_pad_kernel");
  // every frame of every utterance is a row of ONE operand with pitch = hop (rows overlap): C = frames x basis
  int rc = ds2_gemm_f32(0, 0, B * R, 2 * nb, n_fft, ypad, hop, 0, basis, 2 * nb, 0, C, 2 * nb, 0, nullptr, 0, 1, 1, nullptr, 0, stream);
  if (rc) return rc;
  dim3 grid(ceil_div(T, 64), ceil_div(nb, 32), B);
  hipLaunchKernelGGL(stft_post_kernel, grid, dim3(256), 0, s, (const float*)C, 2 * nb, R, nb, T, hop, n_samples_dev, out, part);
  DS2_LAUNCH_CHECK("stft_post_kernel");
  if (normalize) {
    hipLaunchKernelGGL(stft_stats_kernel, dim3(B), dim3(64), 0, s, (const float*)part, (int)(grid.x * grid.y), nb, T, hop, n_samples_dev, stats);
    DS2_LAUNCH_CHECK("stft_stats_kernel");
    const long long total = (long long)B * nb * T;
    int nblk = (int)((total + 255) / 256);
    if (nblk > 16384) nblk = 16384;
    hipLaunchKernelGGL(stft_normalize_kernel, dim3(nblk), dim3(256), 0, s, out, nb, T, hop, n_samples_dev, (const float*)stats, total);
    DS2_LAUNCH_CHECK("stft_normalize_kernel");
  }
  return 0;
}
