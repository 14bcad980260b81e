// FFT kernels of the synthesis path (IR partition spectra, signal block spectra, block synthesis).
// Compiled WITHOUT SLP vectorisation: hipcc's v_pk_*_f32 packing of the butterfly arithmetic makes the in-LDS
// transform 10-25 % slower on gfx950 (profiles/r01_fft_probe.txt), while the spectral MAC in al_kernels.hip
// gains from it; hence two translation units.
#include <hip/hip_runtime.h>
#include <math.h>

#include "al_common.h"
#include "al_fft.h"
#include "al_synth_store.h"

namespace al {

// Kernels are templates on <LOG2M, E>: block of M = 2^LOG2M complex points (= 2M real samples), E complex values
// per thread (al_fft.h).  The launchers pick E = 32 from M = 8192 up unless AL_FLAG_NARROW_FFT asks for 16.
constexpr int min_waves(int e) { return e == 32 ? 2 : 4; }  // register budget: 256 / 128 VGPRs per lane

// ------------------------------------------------------------------ 1. IR partition spectra
// The first half of a partition's 2B window (the B samples themselves) as H float2 per thread; the second half is
// zero.  Loads are unconditional (clamped addresses + selects) so all of them are in flight together.
template <class G>
__device__ __forceinline__ void load_partition(const float *__restrict__ src, int remaining, int tid, float2 (&x)[G::H]) {
  if (remaining >= G::M) {  // workgroup-uniform
#pragma unroll
    for (int m = 0; m < G::H; ++m) x[m] = stream_load<8>(reinterpret_cast<const float2 *>(src + 2 * (tid + G::T * m)));
  } else {
    const int last = remaining - 1;
#pragma unroll
    for (int m = 0; m < G::H; ++m) {
      const int t = 2 * (tid + G::T * m);
      const float x0 = src[min(t, last)], x1 = src[min(t + 1, last)];
      x[m] = make_float2(t <= last ? x0 : 0.f, t + 1 <= last ? x1 : 0.f);
    }
  }
}

// `nb` consecutive partitions of one (emitter, capsule) IR per workgroup; the samples of partition p+1 are
// requested before partition p is transformed, so their HBM latency runs under the transform (this kernel moves
// little data and, with two resident workgroups per CU, would otherwise wait for every load and store).
template <int LOG2M, int E>
__global__ __launch_bounds__((FftGeom<LOG2M, E>::T), min_waves(E)) void k_ir_spectra(al_batch b, int nb) {
  using G = FftGeom<LOG2M, E>;
  constexpr int M = G::M, T = G::T, H = G::H;
  __shared__ float2 s[G::LDS_ELEMS];
  __shared__ float red[48];
  const int tid0 = threadIdx.x;
  const int p0 = blockIdx.x * nb, c = blockIdx.y, n = b.emitter0 + blockIdx.z;
  const int p1 = min(p0 + nb, b.n_partitions);
  FftTwiddles<G> tw;  // requested first: nothing in the loop waits on the table again
  load_fft_twiddles<G, -1>(tw, reinterpret_cast<const float2 *>(b.twiddle), tid0);
  const float *ir = b.ir + (int64_t)c * b.ir_stride_c + (int64_t)n * b.ir_stride_n;
  float2 nxt[H];
  load_partition<G>(ir + (int64_t)p0 * M, b.ir_len - p0 * M, tid0, nxt);
  for (int p = p0; p < p1; ++p) {
    const int tid = opaque_lane(tid0);  // see al_common.h
    tw.hide_from_hoisting();
    float2 v[E];
    float energy = 0.f;
#pragma unroll
    for (int m = 0; m < H; ++m) {
      v[m] = nxt[m];
      energy = fmaf(v[m].x, v[m].x, energy);
      energy = fmaf(v[m].y, v[m].y, energy);
    }
#pragma unroll
    for (int m = H; m < E; ++m) v[m] = make_float2(0.f, 0.f);
    if (p + 1 < p1) load_partition<G>(ir + (int64_t)(p + 1) * M, b.ir_len - (p + 1) * M, tid, nxt);
    const int64_t blk = ((int64_t)n * b.n_capsules + c) * b.n_partitions + p;  // global (energy partials)
    if (b.emitter_parts && b.n_partitions <= AL_SPARSE_MAX_PARTITIONS && p >= b.emitter_parts[n]) {   // workgroup-uniform: energy only
      float mx = 0.f, z = 0.f;
      block_reduce3(energy, mx, z, red, tid, T);
      if (tid == 0) b.ir_energy[blk] = energy;
      __syncthreads();
      continue;
    }

    fft_regs_to_regs<G, -1>(v, s, tw, tid);
    const int64_t hblk = ((int64_t)blockIdx.z * b.n_capsules + c) * b.n_partitions + p;  // chunk-local spectrum
    real_unpack_store_regs<G>(v, s, tw.w0, tid, reinterpret_cast<float2 *>(b.hspec) + hblk * M);
    __syncthreads();  // LDS image and `red` are reused below and by the next partition

    float mx = 0.f, z = 0.f;
    block_reduce3(energy, mx, z, red, tid, T);
    if (tid == 0) b.ir_energy[blk] = energy;
    __syncthreads();
  }
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

template <int LOG2M, int E>
__global__ __launch_bounds__((FftGeom<LOG2M, E>::T), min_waves(E)) void k_signal_spectra(al_batch b) {
  using G = FftGeom<LOG2M, E>;
  constexpr int M = G::M, T = G::T;
  __shared__ float2 s[G::LDS_ELEMS];
  const int tid = threadIdx.x;
  const al_stream st = b.streams[b.stream0 + blockIdx.y];
  if ((int)blockIdx.x >= st.n_j) return;
  const al_event ev = b.events[st.event];
  FftTwiddles<G> tw;
  load_fft_twiddles<G, -1>(tw, reinterpret_cast<const float2 *>(b.twiddle), tid);
  const float *a = b.audio + ev.audio_off;
  const bool moving = st.w_off >= 0 && st.w_len > 0;
  const float *w = b.wtab + (moving ? st.w_off : 0);
  const float clip_gain = st.gain * (b.clip_scale ? b.clip_scale[st.event] : 1.f);  // A13 scale computed on the device
  const int j = st.j_lo + blockIdx.x;
  const int t0 = (j - 1) * M;  // window [(j-1)B, (j+1)B)
  const int last = ev.len - 1;
  float2 v[E];
  if (t0 >= 0 && t0 + 2 * M <= ev.len) {  // interior window (workgroup-uniform): aligned pair loads
#pragma unroll
    for (int m = 0; m < E; ++m) v[m] = *reinterpret_cast<const float2 *>(a + t0 + 2 * (tid + T * m));
  } else {
#pragma unroll
    for (int m = 0; m < E; ++m) {
      const int t = t0 + 2 * (tid + T * m);
      const float x0 = a[min(max(t, 0), last)], x1 = a[min(max(t + 1, 0), last)];
      v[m] = make_float2((t >= 0 && t <= last) ? x0 : 0.f, (t + 1 >= 0 && t + 1 <= last) ? x1 : 0.f);
    }
  }
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const int t = t0 + 2 * (tid + T * m);
    float g0 = clip_gain, g1 = clip_gain;
    if (moving) {  // workgroup-uniform
      g0 *= stream_envelope(w, st.w_len, b.hop, max(t, 0));
      g1 *= stream_envelope(w, st.w_len, b.hop, max(t + 1, 0));
    }
    v[m].x *= g0;
    v[m].y *= g1;
  }
  fft_regs_to_regs<G, -1>(v, s, tw, tid);
  real_unpack_store_regs<G>(v, s, tw.w0, tid, reinterpret_cast<float2 *>(b.xspec) + (int64_t)(st.xspec_base - b.xspec_block0 + blockIdx.x) * M);
}

// ------------------------------------------------------------------ 5. block synthesis
// `nb` consecutive blocks of one (event, capsule) per workgroup.  nb > 1 amortises the workgroup start-up (kernel
// arguments -> event record -> first spectrum is a chain of dependent memory latencies that two resident
// workgroups per CU cannot hide) and lets block k+1 be pulled into this XCD's L2 while block k is transformed.
template <int LOG2M, int E>
__global__ __launch_bounds__((FftGeom<LOG2M, E>::T), min_waves(E)) void k_block_synthesis(al_batch b, int nb) {
  using G = FftGeom<LOG2M, E>;
  constexpr int M = G::M, T = G::T, H = G::H;
  __shared__ float2 s[G::LDS_ELEMS];
  __shared__ float red[48];
  const int tid0 = threadIdx.x;
  const int k0 = blockIdx.x * nb, c = blockIdx.y;
  const al_event ev = b.events[b.event0 + blockIdx.z];
  if (k0 >= ev.n_blocks) return;
  const int k1 = min(k0 + nb, ev.n_blocks);
  FftTwiddles<G> tw;
  load_fft_twiddles<G, 1>(tw, reinterpret_cast<const float2 *>(b.twiddle), tid0);
  float *out = b.spatial + ev.out_off + (int64_t)c * ev.len;
  const float2 *y = reinterpret_cast<const float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + (int64_t)c * ev.n_blocks) * M;
  const bool conv = ev.n_streams > 0;
  const bool pair_ok = (((ev.out_off + (int64_t)c * ev.len) & 1) == 0);

  for (int k = k0; k < k1; ++k) {
    const int tid = opaque_lane(tid0);
    tw.hide_from_hoisting();
    const int tbase = k * M;
    float asum = 0.f, amax = 0.f, bad = 0.f;
    if (!conv) {
      // no emitters: the clip is tiled over the capsules (synthesize.py:572-577)
      const float gain = b.streams[ev.stream0].gain * (b.clip_scale ? b.clip_scale[b.event0 + blockIdx.z] : 1.f);
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
      float2 v[E];
      real_pack_load_regs<G>(y + (int64_t)k * M, v, s, tw.w0, tid, 1.0f / (float)M);
      // one dword per E*8 bytes of the next block (every line for E = 16, every other one for E = 32), into L2
      float touch = 0.f;
      if (k + 1 < k1) touch = reinterpret_cast<const float *>(y + (int64_t)(k + 1) * M)[2 * E * tid];
      fft_regs_to_regs<G, 1>(v, s, tw, tid);
      // keep the alias-free second half of the 2B window: z[n], n in [M/2, M) = samples [B, 2B); those are
      // this thread's registers v[E/2..E-1] (n = tid + T*m), so the result never goes back through LDS
      synth_store_block<G>(v, out, ev, tbase, pair_ok, tid, asum, amax);
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

}  // namespace al
#include "al_split.h"
#include "al_quad16.h"
namespace al {

// ------------------------------------------------------------------ launchers
#define AL_DISPATCH_SPLIT(b, KERNEL, GRID, ...)                                                                          \
  do {                                                                                                                  \
    switch ((b)->log2_block) {                                                                                          \
      case 11: hipLaunchKernelGGL((KERNEL<11>), GRID, dim3(FftGeom<10, 16>::T), 0, stream, __VA_ARGS__); break;         \
      case 12: hipLaunchKernelGGL((KERNEL<12>), GRID, dim3(FftGeom<11, 16>::T), 0, stream, __VA_ARGS__); break;         \
      case 13: hipLaunchKernelGGL((KERNEL<13>), GRID, dim3(FftGeom<12, 16>::T), 0, stream, __VA_ARGS__); break;         \
      case 14: hipLaunchKernelGGL((KERNEL<14>), GRID, dim3(FftGeom<13, 16>::T), 0, stream, __VA_ARGS__); break;         \
      default: return hipErrorInvalidValue;                                                                             \
    }                                                                                                                   \
  } while (0)
static bool use_split(const al_batch *b) { return (b->flags & AL_FLAG_SPLIT_SPECTRA) && b->log2_block >= 11; }
// B = 16384 in four 4096-point tiles (al_quad16.h): the layout of a block IS the quad layout, so the flag selects the kernels
static bool use_quad16(const al_batch *b) { return use_split(b) && (b->flags & AL_FLAG_QUAD_SPECTRA) && b->log2_block == 14; }

#define AL_DISPATCH_GEOM(b, KERNEL, GRID, ...)                                                                        \
  do {                                                                                                                \
    const bool wide = (b)->log2_block >= 13 && !((b)->flags & AL_FLAG_NARROW_FFT);                                      \
    switch ((b)->log2_block) {                                                                                        \
      case 10: hipLaunchKernelGGL((KERNEL<10, 16>), GRID, dim3(FftGeom<10, 16>::T), 0, stream, __VA_ARGS__); break;   \
      case 11: hipLaunchKernelGGL((KERNEL<11, 16>), GRID, dim3(FftGeom<11, 16>::T), 0, stream, __VA_ARGS__); break;   \
      case 12: hipLaunchKernelGGL((KERNEL<12, 16>), GRID, dim3(FftGeom<12, 16>::T), 0, stream, __VA_ARGS__); break;   \
      case 13:                                                                                                        \
        if (wide) hipLaunchKernelGGL((KERNEL<13, 32>), GRID, dim3(FftGeom<13, 32>::T), 0, stream, __VA_ARGS__);       \
        else hipLaunchKernelGGL((KERNEL<13, 16>), GRID, dim3(FftGeom<13, 16>::T), 0, stream, __VA_ARGS__);            \
        break;                                                                                                        \
      case 14:                                                                                                        \
        if (wide) hipLaunchKernelGGL((KERNEL<14, 32>), GRID, dim3(FftGeom<14, 32>::T), 0, stream, __VA_ARGS__);       \
        else hipLaunchKernelGGL((KERNEL<14, 16>), GRID, dim3(FftGeom<14, 16>::T), 0, stream, __VA_ARGS__);            \
        break;                                                                                                        \
      default: return hipErrorInvalidValue;                                                                           \
    }                                                                                                                 \
  } while (0)

hipError_t launch_ir_spectra(const al_batch *b, hipStream_t stream) {
  if (use_quad16(b)) {
    const int run_len = quad16_run_len(b);
    hipLaunchKernelGGL(k_ir_spectra_quad16, dim3(quad16_runs(b, run_len), b->n_capsules, b->n_emitters), dim3(Quad16::T), 0, stream, *b, run_len);
    return hipGetLastError();
  }
  if (use_split(b)) {
    const dim3 grid(b->n_partitions, b->n_capsules, b->n_emitters);
    AL_DISPATCH_SPLIT(b, k_ir_spectra_split, grid, *b);
    return hipGetLastError();
  }
  const int nb = max(1, (b->flags >> 24) & 0x7f);  // AL_FLAG_IR_RUN(n): partitions per workgroup
  const dim3 grid((b->n_partitions + nb - 1) / nb, b->n_capsules, b->n_emitters);
  AL_DISPATCH_GEOM(b, k_ir_spectra, grid, *b, nb);
  return hipGetLastError();
}

// IR + signal spectra in one launch where the split layout makes them the same shape of job; else two launches.
hipError_t launch_forward_spectra(const al_batch *b, hipStream_t stream) {
  const bool have_ir = b->n_emitters > 0, have_sig = b->n_streams > 0 && b->max_nj > 0;
  if (use_split(b) && have_ir && have_sig) {
    const int64_t n_sig = (int64_t)b->max_nj * b->n_streams;
    const int64_t n_ir = (int64_t)b->n_partitions * b->n_capsules * b->n_emitters;
    if (use_quad16(b)) {   // a workgroup per signal window and per run of IR partitions
      const int run_len = quad16_run_len(b), n_runs = quad16_runs(b, run_len);
      const int64_t n_run = (int64_t)n_runs * b->n_capsules * b->n_emitters;
      if (n_sig + n_run <= 0x7fffffff) {
        hipLaunchKernelGGL(k_forward_spectra_quad16, dim3((unsigned)(n_sig + n_run)), dim3(Quad16::T), 0, stream, *b, (int)n_sig, n_runs, run_len);
        return hipGetLastError();
      }
    } else if (n_sig + n_ir <= 0x7fffffff) {
      const dim3 grid((unsigned)(n_sig + n_ir));
      AL_DISPATCH_SPLIT(b, k_forward_spectra_split, grid, *b, (int)n_sig);
      return hipGetLastError();
    }
  }
  if (have_ir)
    if (hipError_t e = launch_ir_spectra(b, stream)) return e;
  if (have_sig) return launch_signal_spectra(b, stream);
  return hipSuccess;
}

hipError_t launch_signal_spectra(const al_batch *b, hipStream_t stream) {
  const dim3 grid(b->max_nj, b->n_streams);
  if (use_quad16(b)) {
    hipLaunchKernelGGL(k_signal_spectra_quad16, grid, dim3(Quad16::T), 0, stream, *b);
    return hipGetLastError();
  }
  if (use_split(b)) {
    AL_DISPATCH_SPLIT(b, k_signal_spectra_split, grid, *b);
    return hipGetLastError();
  }
  AL_DISPATCH_GEOM(b, k_signal_spectra, grid, *b);
  return hipGetLastError();
}

hipError_t launch_block_synthesis(const al_batch *b, hipStream_t stream) {
  if (use_quad16(b)) {
    hipLaunchKernelGGL(k_block_synthesis_quad16, dim3(b->max_blocks, b->n_capsules, b->n_events), dim3(Quad16::T), 0, stream, *b);
    return hipGetLastError();
  }
  if (use_split(b)) {
    const dim3 grid(b->max_blocks, b->n_capsules, b->n_events);
    AL_DISPATCH_SPLIT(b, k_block_synthesis_split, grid, *b);
    return hipGetLastError();
  }
  const int nb = max(1, (b->flags >> 16) & 0xff);  // AL_FLAG_SYNTH_RUN(n): blocks per workgroup
  const dim3 grid((b->max_blocks + nb - 1) / nb, b->n_capsules, b->n_events);
  AL_DISPATCH_GEOM(b, k_block_synthesis, grid, *b, nb);
  return hipGetLastError();
}

}  // namespace al
