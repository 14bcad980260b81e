// Arbitrary-length inverse real FFT in global memory, for the ambience noise synthesis (A12).
//
// irfft of length n over `rows` independent series.  Even n: the half-length trick (one complex
// transform of n/2 points); odd n: a full complex transform of the Hermitian extension.  The complex
// transform is a Stockham auto-sort with radix 4/2/3/5/7 passes through HBM (ping-pong buffers);
// a length with a prime factor above 7 goes through Bluestein's chirp-z on a power-of-two length.
// Twiddles are evaluated in float64 (sincospi) so the float32 result is accurate to ~1e-6 at
// n = 3e6.  This runs once per scene ambience: it is bandwidth-trivial next to the convolution.
#pragma once
#include <hip/hip_runtime.h>

#include "al_fft.h"
#include "al_rng.h"

namespace al {

__device__ __forceinline__ float2 unit_phase(double turns) {  // exp(2*pi*i*turns)
  double s, c;
  sincospi(2.0 * turns, &s, &c);
  return make_float2((float)c, (float)s);
}

// One Stockham pass of radix R over `rows` series of n complex points (inverse: e^{+i...}).
//   out[(j - k) * R + k + q * ns] = sum_r in[j + r * n / R] * w^(r k) * wR^(r q),  k = j mod ns
template <int R>
__global__ __launch_bounds__(256) void k_big_pass(const float2 *__restrict__ in, float2 *__restrict__ out, int64_t n,
                                                  int64_t ns, int dir) {
  const int64_t nb = n / R;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= nb) return;
  const float2 *src = in + (int64_t)blockIdx.y * n;
  float2 *dst = out + (int64_t)blockIdx.y * n;
  const int64_t k = j % ns;
  float2 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = src[j + r * nb];
  if (ns > 1) {
    const double base = (double)dir * (double)k / (double)(ns * R);
#pragma unroll
    for (int r = 1; r < R; ++r) v[r] = cmul(v[r], unit_phase(base * r));
  }
  float2 u[R];
  if (R == 2) {
    u[0] = cadd(v[0], v[1]);
    u[1] = csub(v[0], v[1]);
  } else if (R == 4) {
    if (dir > 0) bfly4<1>(v[0], v[1], v[2], v[3]); else bfly4<-1>(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int q = 0; q < R; ++q) u[q] = v[q];
  } else {
#pragma unroll
    for (int q = 0; q < R; ++q) {
      float2 acc = v[0];
#pragma unroll
      for (int r = 1; r < R; ++r) acc = cadd(acc, cmul(v[r], unit_phase((double)dir * (double)((r * q) % R) / (double)R)));
      u[q] = acc;
    }
  }
  const int64_t base_out = (j - k) * R + k;
#pragma unroll
  for (int q = 0; q < R; ++q) dst[base_out + q * ns] = u[q];
}

// noise spectrum S[f] = shape[f] * (zr + i zi)[f] with the real DC / Nyquist fix-ups (ambience.py:358-369),
// written as the input of the complex inverse transform.
//   even n (m = n/2): Z[k] = E + iO,  E = (S[k] + conj S[m-k])/2,  O = (S[k] - conj S[m-k])/2 * e^{+2 pi i k/n}
//   odd n: the Hermitian extension full[k], k < n.
// Where the standard-normal draws of a spectrum row come from: arrays the host filled (numpy's default_rng: bit-identical
// to the reference for a seed) or the device generator of al_rng.h (zr == nullptr).
struct NoiseDraws {
  const float *zr, *zi;   // this row's draws (host mode)
  uint64_t seed;          // device mode
  int64_t row, bins;
  __device__ __forceinline__ float2 at(int64_t f) const {
    return zr ? make_float2(zr[f], zi[f]) : spectrum_draw(seed, row, bins, f);
  }
};

__device__ __forceinline__ float2 noise_bin(const NoiseDraws &d, const float *shape, int64_t f, int64_t n) {
  const float s = shape ? shape[f] : 1.0f;
  const float2 z = d.at(f);
  float re = z.x * s, im = z.y * s;
  if (f == 0 || (2 * f == n)) {
    im = 0.f;
    re *= 1.41421356237309504880f;
  }
  return make_float2(re, im);
}

__global__ __launch_bounds__(256) void k_noise_pack(const float *zr, const float *zi, uint64_t seed, const float *shape, int64_t n,
                                                    float2 *z) {
  const int64_t bins = n / 2 + 1;
  const NoiseDraws d{zr ? zr + (int64_t)blockIdx.y * bins : nullptr, zi ? zi + (int64_t)blockIdx.y * bins : nullptr, seed,
                     (int64_t)blockIdx.y, bins};
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if ((n & 1) == 0) {
    const int64_t m = n / 2;
    if (k >= m) return;
    const float2 a = noise_bin(d, shape, k, n), b = noise_bin(d, shape, m - k, n);
    const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    const float2 dd = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));
    const float2 o = cmul(dd, unit_phase((double)k / (double)n));
    z[(int64_t)blockIdx.y * m + k] = make_float2(e.x - o.y, e.y + o.x);
  } else {
    if (k >= n) return;
    const float2 a = k <= n / 2 ? noise_bin(d, shape, k, n) : cconj(noise_bin(d, shape, n - k, n));
    z[(int64_t)blockIdx.y * n + k] = a;
  }
}

__global__ __launch_bounds__(256) void k_noise_unpack(const float2 *z, int64_t n, float scale, float *out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float *row = out + (int64_t)blockIdx.y * n;
  if ((n & 1) == 0) {
    const int64_t m = n / 2;
    if (t >= m) return;
    const float2 v = z[(int64_t)blockIdx.y * m + t];
    row[2 * t] = v.x * scale;
    row[2 * t + 1] = v.y * scale;
  } else {
    if (t >= n) return;
    row[t] = z[(int64_t)blockIdx.y * n + t].x * scale;
  }
}

// ---- Bluestein pieces: X[m] = c[m] * sum_k (x[k] c[k]) conj(c)[m-k],  c[k] = exp(dir * i*pi*k^2/n)
__device__ __forceinline__ float2 chirp(int64_t k, int64_t n, int dir) {
  const int64_t q = (k * k) % (2 * n);  // exact phase index
  double s, c;
  sincospi((double)dir * (double)q / (double)n, &s, &c);
  return make_float2((float)c, (float)s);
}

// a[k] = x[k] c[k] (k < n), 0 up to L
__global__ __launch_bounds__(256) void k_blue_pre(const float2 *x, int64_t n, int64_t L, int dir, float2 *a) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= L) return;
  a[(int64_t)blockIdx.y * L + k] = k < n ? cmul(x[(int64_t)blockIdx.y * n + k], chirp(k, n, dir)) : make_float2(0.f, 0.f);
}

// b[j] = conj(c[j]) at j and L-j (one row, shared by all series)
__global__ __launch_bounds__(256) void k_blue_kernel(int64_t n, int64_t L, int dir, float2 *b) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= L) return;
  float2 v = make_float2(0.f, 0.f);
  if (k < n) v = cconj(chirp(k, n, dir));
  else if (L - k < n) v = cconj(chirp(L - k, n, dir));
  b[k] = v;
}

__global__ __launch_bounds__(256) void k_blue_mul(float2 *a, const float2 *b, int64_t L) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= L) return;
  float2 *p = a + (int64_t)blockIdx.y * L + k;
  *p = cmul(*p, b[k]);
}

// x[m] = c[m] * y[m] / L  (m < n)
__global__ __launch_bounds__(256) void k_blue_post(const float2 *y, int64_t n, int64_t L, int dir, float2 *x) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= n) return;
  const float2 v = cmul(y[(int64_t)blockIdx.y * L + m], chirp(m, n, dir));
  const float inv = 1.0f / (float)L;
  x[(int64_t)blockIdx.y * n + m] = make_float2(v.x * inv, v.y * inv);
}

}  // namespace al
