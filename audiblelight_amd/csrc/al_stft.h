// The reference's STFT-domain intermediates of the moving-source path (audiblelight/synthesize.py:109-145 `stft`,
// :184-252 `perform_time_variant_convolution`, :255-274 `istft_overlap_synthesis`) as device kernels, so the public
// functions of the same names exist with the reference's signatures.  The product render path evaluates the identical result in
// the envelope form for the default STFT geometry and its multiples and chains these three on device
// buffers for every other geometry (audiblelight_amd/synthesize.py::_render_moving_general); they are O(frames^2) like the reference.
#pragma once
#include <hip/hip_runtime.h>

#include "al_fft.h"

namespace al {

// z[(row * n_frames + f)][t] = y_padded[row][f*hop + t] * sin^2(pi t / win) for t < win, 0 up to fft_size
// (left pad win - hop, right pad to n_frames*hop: synthesize.py:119-137)
__global__ __launch_bounds__(256) void k_stft_pack(const float *__restrict__ y, int64_t n, int n_frames, int fft_size,
                                                   int win, int hop, int64_t series0, float2 *__restrict__ z) {
  const int64_t series = series0 + blockIdx.y;  // row * n_frames + frame
  const int64_t row = series / n_frames;
  const int f = (int)(series - row * n_frames);
  for (int t = blockIdx.x * 256 + threadIdx.x; t < fft_size; t += gridDim.x * 256) {
    float v = 0.f;
    if (t < win) {
      const int64_t src = (int64_t)f * hop + t - (win - hop);
      if (src >= 0 && src < n) {
        const float s = sinpif((float)t / (float)win);
        v = y[row * n + src] * s * s;
      }
    }
    z[(int64_t)blockIdx.y * fft_size + t] = make_float2(v, 0.f);
  }
}

// spec[series][0 .. fft_size/2] = Z[series][0 .. fft_size/2]
__global__ __launch_bounds__(256) void k_stft_take_half(const float2 *__restrict__ z, int fft_size, int64_t series0,
                                                        float2 *__restrict__ spec) {
  const int n_freq = fft_size / 2 + 1;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n_freq; k += gridDim.x * 256)
    spec[(series0 + blockIdx.y) * n_freq + k] = z[(int64_t)blockIdx.y * fft_size + k];
}

// out[i][f][c] = sum_{k=0}^{min(i, F_ir-1)} S[i-k][f] * sum_l W[i-k][l] * H[k][f][c][l]     (synthesize.py:217-250)
// one thread per (f, c) of one output frame i; layouts as the reference's arrays (C order).
__global__ __launch_bounds__(256) void k_tv_stft_mac(const float2 *__restrict__ s_audio, const float2 *__restrict__ s_ir,
                                                     const float *__restrict__ w, int frame0, int n_frames_ir, int n_freq,
                                                     int n_ch, int n_irs, float2 *__restrict__ out) {
  const int i = frame0 + blockIdx.y;           // launches cover the frame axis in groups of the grid's y limit
  const int fc = blockIdx.x * 256 + threadIdx.x;
  if (fc >= n_freq * n_ch) return;
  const int f = fc / n_ch;
  float2 acc = make_float2(0.f, 0.f);
  const int kmax = min(i, n_frames_ir - 1);
  for (int k = 0; k <= kmax; ++k) {
    const float *wr = w + (int64_t)(i - k) * n_irs;
    const float2 *h = s_ir + ((int64_t)k * n_freq * n_ch + fc) * n_irs;
    float2 ctf = make_float2(0.f, 0.f);
    for (int l = 0; l < n_irs; ++l) {
      const float wl = wr[l];
      if (wl != 0.f) {
        const float2 hv = h[l];
        ctf.x = fmaf(wl, hv.x, ctf.x);
        ctf.y = fmaf(wl, hv.y, ctf.y);
      }
    }
    cfma(acc, s_audio[(int64_t)(i - k) * n_freq + f], ctf);
  }
  out[(int64_t)i * n_freq * n_ch + fc] = acc;
}

// Hermitian extension of spatial_stft[i][0..fft/2][c] into series (i, c) of fft_size complex points (irfft ignores
// the imaginary parts of DC and, for even sizes, Nyquist).
__global__ __launch_bounds__(256) void k_istft_pack(const float2 *__restrict__ spec, int n_freq, int n_ch, int fft_size,
                                                    int64_t series0, float2 *__restrict__ z) {
  const int64_t series = series0 + blockIdx.y;  // i * n_ch + c
  const int64_t i = series / n_ch;
  const int c = (int)(series - i * n_ch);
  for (int k = blockIdx.x * 256 + threadIdx.x; k < fft_size; k += gridDim.x * 256) {
    const int src = k <= fft_size / 2 ? k : fft_size - k;
    float2 v = spec[(i * n_freq + src) * n_ch + c];
    if (k > fft_size / 2) v.y = -v.y;
    if (k == 0 || 2 * k == fft_size) v.y = 0.f;
    z[(int64_t)blockIdx.y * fft_size + k] = v;
  }
}

// frames[(i * n_ch + c)][t] (real part of the unnormalised inverse transform) overlap-added at i*hop, then the slice
// [win, n_frames*hop) (synthesize.py:266-274).  Gather form: each output sample sums the frames that cover it.
__global__ __launch_bounds__(256) void k_istft_ola(const float2 *__restrict__ frames, int n_frames, int n_ch, int fft_size,
                                                   int win, int hop, float *__restrict__ out) {
  const int64_t n_out = (int64_t)n_frames * hop - win;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_out * n_ch) return;
  const int64_t t = idx / n_ch + win;
  const int c = (int)(idx % n_ch);
  int64_t i_hi = t / hop;
  if (i_hi > n_frames - 1) i_hi = n_frames - 1;
  int64_t i_lo = (t - fft_size + hop) / hop;  // smallest i with t - i*hop < fft_size
  if (i_lo < 0) i_lo = 0;
  float acc = 0.f;
  for (int64_t i = i_lo; i <= i_hi; ++i) {
    const int64_t off = t - i * hop;
    if (off >= 0 && off < fft_size) acc += frames[(i * n_ch + c) * fft_size + off].x;
  }
  out[idx] = acc;
}

}  // namespace al
