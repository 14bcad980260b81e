// FFT kernels of the synthesis path (IR partition spectra, signal block spectra, block synthesis).
// Compiled WITHOUT SLP vectorisation: hipcc's v_pk_*_f32 packing of the butterfly arithmetic makes the in-LDS
// transform 10-25 % slower on gfx950 (profiles/r01_fft_probe.txt), while the spectral MAC in al_kernels.hip
// gains from it; hence two translation units.
#include <hip/hip_runtime.h>
#include <math.h>

#include "al_common.h"
#include "al_fft.h"

namespace al {

// ------------------------------------------------------------------ 1. IR partition spectra
template <int LOG2M>
__global__ __launch_bounds__(fft_threads(LOG2M), 4) void k_ir_spectra(al_batch b) {
  constexpr int M = 1 << LOG2M, T = M / 16;
  __shared__ float2 s[fft_lds_elems(LOG2M)];
  __shared__ float red[48];
  const int tid = threadIdx.x;
  const int p = blockIdx.x, c = blockIdx.y, n = b.emitter0 + blockIdx.z;
  const float2 *tw = reinterpret_cast<const float2 *>(b.twiddle);
  const float *src = b.ir + (int64_t)c * b.ir_stride_c + (int64_t)n * b.ir_stride_n + (int64_t)p * M;
  const int remaining = b.ir_len - p * M;  // samples of this partition that exist (may exceed M)
  float2 v[16];
  float energy = 0.f;
  // first half of the 2B window = the partition, second half zero.  Loads are unconditional
  // (clamped addresses + selects) so all eight are in flight together.
  if (remaining >= M) {  // workgroup-uniform
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = *reinterpret_cast<const float2 *>(src + 2 * (tid + T * m));
  } else {
    const int last = remaining - 1;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int t = 2 * (tid + T * m);
      const float x0 = src[min(t, last)], x1 = src[min(t + 1, last)];
      v[m] = make_float2(t <= last ? x0 : 0.f, t + 1 <= last ? x1 : 0.f);
    }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    energy = fmaf(v[m].x, v[m].x, energy);
    energy = fmaf(v[m].y, v[m].y, energy);
  }
#pragma unroll
  for (int m = 8; m < 16; ++m) v[m] = make_float2(0.f, 0.f);

  fft_regs_to_regs<LOG2M, -1>(v, s, tw, tid);
  const int64_t blk = ((int64_t)n * b.n_capsules + c) * b.n_partitions + p;  // global (energy partials)
  const int64_t hblk = ((int64_t)blockIdx.z * b.n_capsules + c) * b.n_partitions + p;  // chunk-local spectrum
  real_unpack_store_regs<LOG2M>(v, s, tw, tid, reinterpret_cast<float2 *>(b.hspec) + hblk * M);
  __syncthreads();  // the reduction below reuses LDS-adjacent scratch only, but keep phases separate

  float mx = 0.f, z = 0.f;
  block_reduce3(energy, mx, z, red, tid, T);
  if (tid == 0) b.ir_energy[blk] = energy;
}

// Split variant (see al_fft.h "Split transforms"): M/32 threads, two half-size transforms, half the LDS.
template <int LOG2M>
__global__ __launch_bounds__(fft_threads(LOG2M - 1)) void k_ir_spectra_split(al_batch b) {
  constexpr int M = 1 << LOG2M, MH = M / 2, T = MH / 16;
  __shared__ float2 s[fft_lds_elems(LOG2M - 1)];
  __shared__ float red[48];
  const int tid = threadIdx.x;
  const int p = blockIdx.x, c = blockIdx.y, n = b.emitter0 + blockIdx.z;
  const float2 *tw = reinterpret_cast<const float2 *>(b.twiddle);
  const float *src = b.ir + (int64_t)c * b.ir_stride_c + (int64_t)n * b.ir_stride_n + (int64_t)p * M;
  const int remaining = b.ir_len - p * M;
  // the partition fills complex points [0, MH) of the window, the upper half is zero: a[n] = z[n], b[n] = z[n] w^n
  float2 z[16];
  if (remaining >= M) {
#pragma unroll
    for (int m = 0; m < 16; ++m) z[m] = *reinterpret_cast<const float2 *>(src + 2 * (tid + T * m));
  } else {
    const int last = remaining - 1;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int t = 2 * (tid + T * m);
      const float x0 = src[min(t, last)], x1 = src[min(t + 1, last)];
      z[m] = make_float2(t <= last ? x0 : 0.f, t + 1 <= last ? x1 : 0.f);
    }
  }
  float energy = 0.f;
  float2 v[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    energy = fmaf(z[m].x, z[m].x, energy);
    energy = fmaf(z[m].y, z[m].y, energy);
    v[m] = z[m];
  }
  const int64_t blk = ((int64_t)n * b.n_capsules + c) * b.n_partitions + p;
  const int64_t hblk = ((int64_t)blockIdx.z * b.n_capsules + c) * b.n_partitions + p;
  float2 *out = reinterpret_cast<float2 *>(b.hspec) + hblk * M;
  fft_regs_to_regs<LOG2M - 1, -1, 2>(v, s, tw, tid);
  split_unpack_store_even<LOG2M>(v, s, tw, tid, out);
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = cmul(z[m], tw[2 * (tid + T * m)]);
  fft_regs_to_regs<LOG2M - 1, -1, 2>(v, s, tw, tid);
  split_unpack_store_odd<LOG2M>(v, s, tw, tid, out);
  float mx = 0.f, zz = 0.f;
  block_reduce3(energy, mx, zz, red, tid, T);
  if (tid == 0) b.ir_energy[blk] = energy;
}

// ------------------------------------------------------------------ 3. signal block spectra
// Cross-fade envelope of a moving stream at sample t:
//   env(t) = W[q+1] * win(r) + W[q] * (1 - win(r)),  q = t / hop, r = t % hop, win(r) = sin^2(pi r / (2 hop))
__device__ __forceinline__ float stream_envelope(const float *__restrict__ w, int w_len, int hop, int t) {
  const int q = t / hop, r = t - q * hop;
  const float sn = sinpif((float)r / (float)(2 * hop));
  const float win = sn * sn;
  const float a0 = w[min(q, w_len - 1)], a1 = w[min(q + 1, w_len - 1)];  // unconditional loads
  const float w0 = q < w_len ? a0 : 0.f;
  const float w1 = q + 1 < w_len ? a1 : 0.f;
  return fmaf(w1 - w0, win, w0);
}

template <int LOG2M>
__global__ __launch_bounds__(fft_threads(LOG2M), 4) void k_signal_spectra(al_batch b) {
  constexpr int M = 1 << LOG2M, T = M / 16;
  __shared__ float2 s[fft_lds_elems(LOG2M)];
  const int tid = threadIdx.x;
  const al_stream st = b.streams[b.stream0 + blockIdx.y];
  if ((int)blockIdx.x >= st.n_j) return;
  const al_event ev = b.events[st.event];
  const float2 *tw = reinterpret_cast<const float2 *>(b.twiddle);
  const float *a = b.audio + ev.audio_off;
  const bool moving = st.w_off >= 0 && st.w_len > 0;
  const float *w = b.wtab + (moving ? st.w_off : 0);
  const int j = st.j_lo + blockIdx.x;
  const int t0 = (j - 1) * M;  // window [(j-1)B, (j+1)B)
  const int last = ev.len - 1;
  float2 v[16];
  if (t0 >= 0 && t0 + 2 * M <= ev.len) {  // interior window (workgroup-uniform): aligned pair loads
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = *reinterpret_cast<const float2 *>(a + t0 + 2 * (tid + T * m));
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int t = t0 + 2 * (tid + T * m);
      const float x0 = a[min(max(t, 0), last)], x1 = a[min(max(t + 1, 0), last)];
      v[m] = make_float2((t >= 0 && t <= last) ? x0 : 0.f, (t + 1 >= 0 && t + 1 <= last) ? x1 : 0.f);
    }
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int t = t0 + 2 * (tid + T * m);
    float g0 = st.gain, g1 = st.gain;
    if (moving) {  // workgroup-uniform
      g0 *= stream_envelope(w, st.w_len, b.hop, max(t, 0));
      g1 *= stream_envelope(w, st.w_len, b.hop, max(t + 1, 0));
    }
    v[m].x *= g0;
    v[m].y *= g1;
  }
  fft_regs_to_regs<LOG2M, -1>(v, s, tw, tid);
  real_unpack_store_regs<LOG2M>(v, s, tw, tid, reinterpret_cast<float2 *>(b.xspec) + (int64_t)(st.xspec_base - b.xspec_block0 + blockIdx.x) * M);
}

template <int LOG2M>
__global__ __launch_bounds__(fft_threads(LOG2M - 1)) void k_signal_spectra_split(al_batch b) {
  constexpr int M = 1 << LOG2M, MH = M / 2, T = MH / 16;
  __shared__ float2 s[fft_lds_elems(LOG2M - 1)];
  const int tid = threadIdx.x;
  const al_stream st = b.streams[b.stream0 + blockIdx.y];
  if ((int)blockIdx.x >= st.n_j) return;
  const al_event ev = b.events[st.event];
  const float2 *tw = reinterpret_cast<const float2 *>(b.twiddle);
  const float *a = b.audio + ev.audio_off;
  const bool moving = st.w_off >= 0 && st.w_len > 0;
  const float *w = b.wtab + (moving ? st.w_off : 0);
  const int j = st.j_lo + blockIdx.x;
  const int t0 = (j - 1) * M;  // window [(j-1)B, (j+1)B) = complex points [0, M)
  const int last = ev.len - 1;
  const bool interior = (t0 >= 0 && t0 + 2 * M <= ev.len);
  float2 va[16], vb[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    float2 lo, hi;  // z[n] and z[n + MH]
    const int tl = t0 + 2 * (tid + T * m), th = tl + M;
    if (interior) {
      lo = *reinterpret_cast<const float2 *>(a + tl);
      hi = *reinterpret_cast<const float2 *>(a + th);
    } else {
      const float l0 = a[min(max(tl, 0), last)], l1 = a[min(max(tl + 1, 0), last)];
      const float h0 = a[min(max(th, 0), last)], h1 = a[min(max(th + 1, 0), last)];
      lo = make_float2((tl >= 0 && tl <= last) ? l0 : 0.f, (tl + 1 >= 0 && tl + 1 <= last) ? l1 : 0.f);
      hi = make_float2((th >= 0 && th <= last) ? h0 : 0.f, (th + 1 >= 0 && th + 1 <= last) ? h1 : 0.f);
    }
    float gl0 = st.gain, gl1 = st.gain, gh0 = st.gain, gh1 = st.gain;
    if (moving) {
      gl0 *= stream_envelope(w, st.w_len, b.hop, max(tl, 0));
      gl1 *= stream_envelope(w, st.w_len, b.hop, max(tl + 1, 0));
      gh0 *= stream_envelope(w, st.w_len, b.hop, max(th, 0));
      gh1 *= stream_envelope(w, st.w_len, b.hop, max(th + 1, 0));
    }
    lo.x *= gl0; lo.y *= gl1; hi.x *= gh0; hi.y *= gh1;
    va[m] = cadd(lo, hi);
    vb[m] = cmul(csub(lo, hi), tw[2 * (tid + T * m)]);
  }
  float2 *out = reinterpret_cast<float2 *>(b.xspec) + (int64_t)(st.xspec_base - b.xspec_block0 + blockIdx.x) * M;
  fft_regs_to_regs<LOG2M - 1, -1, 2>(va, s, tw, tid);
  split_unpack_store_even<LOG2M>(va, s, tw, tid, out);
  fft_regs_to_regs<LOG2M - 1, -1, 2>(vb, s, tw, tid);
  split_unpack_store_odd<LOG2M>(vb, s, tw, tid, out);
}

// ------------------------------------------------------------------ 5. block synthesis
// `nb` consecutive blocks of one (event, capsule) per workgroup.  nb > 1 amortises the workgroup start-up (kernel
// arguments -> event record -> first spectrum is a chain of dependent memory latencies that two resident
// workgroups per CU cannot hide) and lets block k+1 be pulled into this XCD's L2 while block k is transformed.
template <int LOG2M>
__global__ __launch_bounds__(fft_threads(LOG2M), 4) void k_block_synthesis(al_batch b, int nb) {
  constexpr int M = 1 << LOG2M, T = M / 16;
  __shared__ float2 s[fft_lds_elems(LOG2M)];
  __shared__ float red[48];
  const int tid0 = threadIdx.x;
  const int k0 = blockIdx.x * nb, c = blockIdx.y;
  const al_event ev = b.events[b.event0 + blockIdx.z];
  if (k0 >= ev.n_blocks) return;
  const int k1 = min(k0 + nb, ev.n_blocks);
  const float2 *tw = reinterpret_cast<const float2 *>(b.twiddle);
  float *out = b.spatial + ev.out_off + (int64_t)c * ev.len;
  const float2 *y = reinterpret_cast<const float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + (int64_t)c * ev.n_blocks) * M;
  const bool conv = ev.n_streams > 0;
  const bool pair_ok = (((ev.out_off + (int64_t)c * ev.len) & 1) == 0);

  for (int k = k0; k < k1; ++k) {
    const int tid = opaque_lane(tid0);
    const int tbase = k * M;
    float asum = 0.f, amax = 0.f, bad = 0.f;
    if (!conv) {
      // no emitters: the clip is tiled over the capsules (synthesize.py:572-577)
      const float gain = b.streams[ev.stream0].gain;
      const float *a = b.audio + ev.audio_off;
      for (int i = tid; i < M; i += T) {
        const int t = tbase + i;
        if (t < ev.len) {
          const float x = a[t] * gain;
          out[t] = x;
          asum += fabsf(x);
          amax = fmaxf(amax, fabsf(x));
          bad += isfinite(x) ? 0.f : 1.f;
        }
      }
    } else {
      float2 v[16];
      real_pack_load_regs<LOG2M>(y + (int64_t)k * M, v, s, tw, tid, 1.0f / (float)M);
      // one dword per 128-byte line of the next block: T threads x 128 B = the whole spectrum, into L2
      float touch = 0.f;
      if (k + 1 < k1) touch = reinterpret_cast<const float *>(y + (int64_t)(k + 1) * M)[32 * tid];
      fft_regs_to_regs<LOG2M, 1>(v, s, tw, tid);
      // keep the alias-free second half of the 2B window: z[n], n in [M/2, M) = samples [B, 2B); those are
      // this thread's registers v[8..15] (n = tid + T*m), so the result never goes back through LDS
      if (pair_ok && tbase + M <= ev.valid_len) {  // interior block (workgroup-uniform): unconditional pair stores
        float *o = out + tbase + 2 * tid;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float2 z = v[8 + m];
          *reinterpret_cast<float2 *>(o + 2 * T * m) = z;
          asum += fabsf(z.x) + fabsf(z.y);
          amax = fmaxf(amax, fmaxf(fabsf(z.x), fabsf(z.y)));
        }
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int i = tid + T * m;  // complex index inside the kept half
          const float2 z = v[8 + m];
          const int t = tbase + 2 * i;
          const float x0 = t < ev.valid_len ? z.x : 0.f;
          const float x1 = t + 1 < ev.valid_len ? z.y : 0.f;
          if (t < ev.len) {
            out[t] = x0;
            asum += fabsf(x0);
            amax = fmaxf(amax, fabsf(x0));
          }
          if (t + 1 < ev.len) {
            out[t + 1] = x1;
            asum += fabsf(x1);
            amax = fmaxf(amax, fabsf(x1));
          }
        }
      }
      // a NaN or Inf anywhere makes the sum of magnitudes non-finite: one test per thread instead of one per sample
      bad = isfinite(asum) ? 0.f : 1.f;
      if (bad != 0.f) { asum = 0.f; amax = 0.f; }
      if (__float_as_uint(touch) == 0x7fc12345u) bad = 1.f;  // keeps the touch load alive; a NaN marks the block anyway
    }
    block_reduce3(asum, amax, bad, red, tid, T);
    if (tid == 0) {
      float *pp = b.partials + 4 * ((int64_t)ev.part_base + (int64_t)c * ev.n_blocks + k);
      pp[0] = asum;
      pp[1] = amax;
      pp[2] = bad;
      pp[3] = 0.f;
    }
    __syncthreads();  // `red` and the LDS image are reused by the next block
  }
}

template <int LOG2M>
__global__ __launch_bounds__(fft_threads(LOG2M - 1)) void k_block_synthesis_split(al_batch b) {
  constexpr int M = 1 << LOG2M, MH = M / 2, T = MH / 16;
  __shared__ float2 s[fft_lds_elems(LOG2M - 1)];
  __shared__ float red[48];
  const int tid = threadIdx.x;
  const int k = blockIdx.x, c = blockIdx.y;
  const al_event ev = b.events[b.event0 + blockIdx.z];
  if (k >= ev.n_blocks) return;
  const float2 *tw = reinterpret_cast<const float2 *>(b.twiddle);
  float *out = b.spatial + ev.out_off + (int64_t)c * ev.len;
  const int tbase = k * M;
  float asum = 0.f, amax = 0.f, bad = 0.f;
  if (ev.n_streams <= 0) {
    const float gain = b.streams[ev.stream0].gain;
    const float *a = b.audio + ev.audio_off;
    for (int i = tid; i < M; i += T) {
      const int t = tbase + i;
      if (t < ev.len) {
        const float x = a[t] * gain;
        out[t] = x;
        asum += fabsf(x);
        amax = fmaxf(amax, fabsf(x));
        bad += isfinite(x) ? 0.f : 1.f;
      }
    }
  } else {
    const float2 *y = reinterpret_cast<const float2 *>(b.yspec) +
                      ((int64_t)(ev.yspec_base - b.yspec_block0) + (int64_t)c * ev.n_blocks + k) * M;
    const float scale = 1.0f / (float)M;
    float2 va[16], vb[16];
    split_pack_load<LOG2M, 0>(y, va, s, tw, tid, scale);
    fft_regs_to_regs<LOG2M - 1, 1, 2>(va, s, tw, tid);
    __syncthreads();
    split_pack_load<LOG2M, 1>(y, vb, s, tw, tid, scale);
    fft_regs_to_regs<LOG2M - 1, 1, 2>(vb, s, tw, tid);
    // the alias-free half of the window: z[n + M/2] = A[n] - conj(w)^n B[n], n = tid + T*m  ->  samples kB + 2n
    const bool pair_ok = (((ev.out_off + (int64_t)c * ev.len) & 1) == 0);
    const bool interior = pair_ok && (tbase + M <= ev.valid_len);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int n = tid + T * m;
      const float2 z = csub(va[m], cmul(cconj(tw[2 * n]), vb[m]));
      const int t = tbase + 2 * n;
      if (interior) {
        *reinterpret_cast<float2 *>(out + t) = z;
        asum += fabsf(z.x) + fabsf(z.y);
        amax = fmaxf(amax, fmaxf(fabsf(z.x), fabsf(z.y)));
        bad += (isfinite(z.x) ? 0.f : 1.f) + (isfinite(z.y) ? 0.f : 1.f);
      } else {
        const float x0 = t < ev.valid_len ? z.x : 0.f;
        const float x1 = t + 1 < ev.valid_len ? z.y : 0.f;
        if (t < ev.len) {
          out[t] = x0;
          asum += fabsf(x0);
          amax = fmaxf(amax, fabsf(x0));
          bad += isfinite(x0) ? 0.f : 1.f;
        }
        if (t + 1 < ev.len) {
          out[t + 1] = x1;
          asum += fabsf(x1);
          amax = fmaxf(amax, fabsf(x1));
          bad += isfinite(x1) ? 0.f : 1.f;
        }
      }
    }
  }
  block_reduce3(asum, amax, bad, red, tid, T);
  if (tid == 0) {
    float *pp = b.partials + 4 * ((int64_t)ev.part_base + (int64_t)c * ev.n_blocks + k);
    pp[0] = asum;
    pp[1] = amax;
    pp[2] = bad;
    pp[3] = 0.f;
  }
}

// ------------------------------------------------------------------ launchers
namespace {
// Blocks of 8192 and 16384 points may use the split transforms (half the LDS image => more resident workgroups);
// measured slower on MI355X because of their half-line global accesses, so they are opt-in (AL_FLAG_FORCE_SPLIT).
bool use_split(const al_batch *b) { return (b->flags & AL_FLAG_FORCE_SPLIT) && b->log2_block >= 12; }
}  // namespace

#define AL_DISPATCH_LOG2(log2, CALL)            \
  switch (log2) {                               \
    case 10: { constexpr int L = 10; CALL; } break; \
    case 11: { constexpr int L = 11; CALL; } break; \
    case 12: { constexpr int L = 12; CALL; } break; \
    case 13: { constexpr int L = 13; CALL; } break; \
    case 14: { constexpr int L = 14; CALL; } break; \
    default: return hipErrorInvalidValue;       \
  }
#define AL_DISPATCH_SPLIT(log2, CALL)           \
  switch (log2) {                               \
    case 12: { constexpr int L = 12; CALL; } break; \
    case 13: { constexpr int L = 13; CALL; } break; \
    case 14: { constexpr int L = 14; CALL; } break; \
    default: return hipErrorInvalidValue;       \
  }

hipError_t launch_ir_spectra(const al_batch *b, hipStream_t stream) {
  const dim3 grid(b->n_partitions, b->n_capsules, b->n_emitters);
  if (use_split(b)) {
    AL_DISPATCH_SPLIT(b->log2_block, hipLaunchKernelGGL((k_ir_spectra_split<L>), grid, dim3(fft_threads(L - 1)), 0, stream, *b));
  } else {
    AL_DISPATCH_LOG2(b->log2_block, hipLaunchKernelGGL((k_ir_spectra<L>), grid, dim3(fft_threads(L)), 0, stream, *b));
  }
  return hipGetLastError();
}

hipError_t launch_signal_spectra(const al_batch *b, hipStream_t stream) {
  const dim3 grid(b->max_nj, b->n_streams);
  if (use_split(b)) {
    AL_DISPATCH_SPLIT(b->log2_block, hipLaunchKernelGGL((k_signal_spectra_split<L>), grid, dim3(fft_threads(L - 1)), 0, stream, *b));
  } else {
    AL_DISPATCH_LOG2(b->log2_block, hipLaunchKernelGGL((k_signal_spectra<L>), grid, dim3(fft_threads(L)), 0, stream, *b));
  }
  return hipGetLastError();
}

hipError_t launch_block_synthesis(const al_batch *b, hipStream_t stream) {
  const dim3 grid(b->max_blocks, b->n_capsules, b->n_events);
  if (use_split(b)) {
    AL_DISPATCH_SPLIT(b->log2_block, hipLaunchKernelGGL((k_block_synthesis_split<L>), grid, dim3(fft_threads(L - 1)), 0, stream, *b));
  } else {
    const int nb = max(1, (b->flags >> 16) & 0xff);  // AL_FLAG_SYNTH_RUN(n): blocks per workgroup
    const dim3 gridn((b->max_blocks + nb - 1) / nb, b->n_capsules, b->n_events);
    AL_DISPATCH_LOG2(b->log2_block, hipLaunchKernelGGL((k_block_synthesis<L>), gridn, dim3(fft_threads(L)), 0, stream, *b, nb));
  }
  return hipGetLastError();
}

}  // namespace al
