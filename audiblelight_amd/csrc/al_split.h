// Split-layout transforms (included by al_transforms.hip): the three FFT kernels of the pipeline at HALF their transform
// size, for B >= 2048.
//
// A block's spectrum (B complex slots for a 2B-sample real window w = [w1, w2]) is stored as two halves:
//   slots [0, B/2):   the EVEN bins W[2f] = rFFT_B(w1 + w2)[f]  (slot 0 packs W[0] and W[B], both real), and
//   slots [B/2, B):   U[q] = W[4q + 1] = sum_n d[n] e^{-2 pi i (2q + 1/2) n / B}, d = w1 - w2: the ODD bins reachable
//                     as the B/2-point FFT of u[n'] = (d[n'] - i d[n' + B/2]) e^{-i pi n'/B}  (the other odd bins are
//                     their conjugate mirrors, W[2B-1-4q] = conj U[q]: no separate storage).
// The spectral accumulate is element-wise and only treats slot 0 specially, so k_spectral_mac* run unchanged on this
// layout; what changes is that every window is transformed as two INDEPENDENT B/2-point complex FFTs instead of one
// B-point FFT: half the LDS per workgroup (35 KB instead of 68 KB at B = 8192: four resident workgroups per CU instead
// of two), three passes of 16 instead of 32 x 16 x 16.  The odd half needs no real-FFT packing step at all.
// Inverse: w2 = (s - d) / 2 with s = irFFT_B(even half), d from u = iFFT_{B/2}(U) * e^{+i pi n'/B}: the alias-free half
// that overlap-save keeps (checked against the one-transform form in tests; formulas: DESIGN.md section 5, "Split layout").
#pragma once
#include <hip/hip_runtime.h>

#include "al_common.h"
#include "al_fft.h"
#include "al_synth_store.h"

#ifndef AL_SPLIT_WAVES
#define AL_SPLIT_WAVES 3   /* minimum waves per SIMD the split kernels are compiled for (register budget) */
#endif

namespace al {

// What synth_store_block needs to know about a split block: B = 2*M samples as E pairs per thread (in the upper half of
// a 2*E register image, like the alias-free half of a full-size transform).
template <class G>
struct SplitOut {
  static constexpr int M = 2 * G::M, T = G::T, H = G::E, E = 2 * G::E;
};

// exp(-i*pi*(tid + T*i)/(2*M)) for i < E from ONE table entry (the table is the block's: 2*M = B points) and
// compile-time factors exp(-i*pi*i/(2*E)).
template <class G>
struct HalfTurnFactors {
  float c[G::E], s[G::E];
  constexpr HalfTurnFactors() : c{}, s{} {
    for (int i = 0; i < G::E; ++i) {
      c[i] = (float)ct_cospi(i, 2 * G::E);
      s[i] = (float)ct_sinpi(i, 2 * G::E);
    }
  }
};

// Window samples in "pair" layout: d[i] = (d[2m], d[2m+1]), m = tid + T*i, of the difference d = w1 - w2 of the window's
// halves.  Returns the odd half's transform input u[n'] = (d[n'] - i d[n' + B/2]) e^{-i pi n'/B}, n' = tid + T*i: d
// changes layout through LDS (B floats: fits the transform image, which is free before the first pass).
template <class G>
__device__ __forceinline__ void split_odd_input(const float2 (&d)[G::E], float2 (&u)[G::E], float2 *s, float2 w8, int tid) {
  constexpr int T = G::T, E = G::E, M = G::M;   // M = B/2 complex points, B = 2*M samples per half window
  constexpr HalfTurnFactors<G> hf{};
  float *dl = reinterpret_cast<float *>(s);
#pragma unroll
  for (int i = 0; i < E; ++i) reinterpret_cast<float2 *>(dl)[tid + T * i] = d[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int n = tid + T * i;
    const float2 v = make_float2(dl[n], -dl[n + M]);                       // d[n'] - i d[n' + B/2]   (B/2 = M)
    u[i] = cmul(v, cmul(w8, make_float2(hf.c[i], -hf.s[i])));              // * exp(-i pi n'/B)
  }
  __syncthreads();   // the image is free again for the transforms
}

// ------------------------------------------------------------------ 1s. IR partition spectra, split layout
template <int LOG2M>
__device__ __forceinline__ void ir_spectra_split_body(const al_batch &b, float2 *s, float *red, int p, int c, int nz) {
  using G = FftGeom<LOG2M - 1, 16>;
  using L = PlainSlots;
  constexpr int M = G::M, T = G::T, E = G::E, B = 2 * M;
  const int tid = threadIdx.x;
  const int n = b.emitter0 + nz;
  const float2 *table = reinterpret_cast<const float2 *>(b.twiddle);
  FftTwiddles<G> tw;
  load_fft_twiddles<G, -1, 2>(tw, table, tid);
  const float2 w8 = table[tid];
  const float *ir = b.ir + (int64_t)c * b.ir_stride_c + (int64_t)n * b.ir_stride_n + (int64_t)p * B;
  const int remaining = b.ir_len - p * B;
  float2 a[E];
  if (remaining >= B) {   // workgroup-uniform
#pragma unroll
    for (int i = 0; i < E; ++i) a[i] = stream_load<8>(reinterpret_cast<const float2 *>(ir + 2 * (tid + T * i)));
  } else {
    const int last = remaining - 1;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int t = 2 * (tid + T * i);
      const float x0 = ir[min(t, last)], x1 = ir[min(t + 1, last)];
      a[i] = make_float2(t <= last ? x0 : 0.f, t + 1 <= last ? x1 : 0.f);
    }
  }
  float energy = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    energy = fmaf(a[i].x, a[i].x, energy);
    energy = fmaf(a[i].y, a[i].y, energy);
  }
  const int64_t blk = ((int64_t)n * b.n_capsules + c) * b.n_partitions + p;             // global (energy partials)
  if (b.emitter_parts && b.n_partitions <= AL_SPARSE_MAX_PARTITIONS && p >= b.emitter_parts[n]) {   // workgroup-uniform: no kept block hears it
    float mx = 0.f, zz = 0.f;                         // (al_batch.emitter_parts); its energy still counts for normalize_irs
    block_reduce3(energy, mx, zz, red, tid, T);
    if (tid == 0) b.ir_energy[blk] = energy;
    return;
  }
  const int64_t hblk = ((int64_t)nz * b.n_capsules + c) * b.n_partitions + p;           // chunk-local spectrum
  float2 *out = reinterpret_cast<float2 *>(b.hspec) + hblk * B;
  {   // odd half first: the second half of the window is zero, so d = s = h and `a` stays for the even half
    float2 u[E];
    split_odd_input<G>(a, u, s, w8, tid);
    fft_regs_to_regs<G, -1>(u, s, tw, tid);
#pragma unroll
    for (int i = 0; i < E; ++i) L::store_odd(out + M, tid + T * i, u[i]);
  }
  fft_regs_to_regs<G, -1>(a, s, tw, tid);
  real_unpack_store_regs<G, L>(a, s, tw.w0, tid, out);
  __syncthreads();   // `red` below shares nothing with the image, but the image's last reads must be over
  float mx = 0.f, zz = 0.f;
  block_reduce3(energy, mx, zz, red, tid, T);
  if (tid == 0) b.ir_energy[blk] = energy;
}

template <int LOG2M>
__global__ __launch_bounds__((FftGeom<LOG2M - 1, 16>::T), AL_SPLIT_WAVES) void k_ir_spectra_split(al_batch b) {
  __shared__ float2 s[FftGeom<LOG2M - 1, 16>::LDS_ELEMS];
  __shared__ float red[48];
  ir_spectra_split_body<LOG2M>(b, s, red, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ------------------------------------------------------------------ 3s. signal block spectra, split layout
template <int LOG2M>
__device__ __forceinline__ void signal_spectra_split_body(const al_batch &b, float2 *s, int jblock, int stream_index) {
  using G = FftGeom<LOG2M - 1, 16>;
  using L = PlainSlots;
  constexpr int M = G::M, T = G::T, E = G::E, B = 2 * M;
  const int tid = threadIdx.x;
  const al_stream st = b.streams[b.stream0 + stream_index];
  if (jblock >= st.n_j) return;
  const al_event ev = b.events[st.event];
  const float2 *table = reinterpret_cast<const float2 *>(b.twiddle);
  FftTwiddles<G> tw;
  load_fft_twiddles<G, -1, 2>(tw, table, tid);
  const float2 w8 = table[tid];
  const float *a = b.audio + ev.audio_off;
  const bool moving = st.w_off >= 0 && st.w_len > 0;
  const float *w = b.wtab + (moving ? st.w_off : 0);
  const float clip_gain = st.gain * (b.clip_scale ? b.clip_scale[st.event] : 1.f);
  const int j = st.j_lo + jblock;
  const int t0 = (j - 1) * B;  // window [(j-1)B, (j+1)B)
  const int last = ev.len - 1;
  float2 h1[E], h2[E];
  auto load_half = [&](float2 (&dst)[E], int base) {
    if (base >= 0 && base + B <= ev.len) {   // interior half window (workgroup-uniform): aligned pair loads
#pragma unroll
      for (int i = 0; i < E; ++i) dst[i] = *reinterpret_cast<const float2 *>(a + base + 2 * (tid + T * i));
    } else {
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int t = base + 2 * (tid + T * i);
        const float x0 = a[min(max(t, 0), last)], x1 = a[min(max(t + 1, 0), last)];
        dst[i] = make_float2((t >= 0 && t <= last) ? x0 : 0.f, (t + 1 >= 0 && t + 1 <= last) ? x1 : 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int t = base + 2 * (tid + T * i);
      float g0 = clip_gain, g1 = clip_gain;
      if (moving) {  // workgroup-uniform
        g0 *= stream_envelope(w, st.w_len, b.hop, max(t, 0));
        g1 *= stream_envelope(w, st.w_len, b.hop, max(t + 1, 0));
      }
      dst[i].x *= g0;
      dst[i].y *= g1;
    }
  };
  load_half(h1, t0);
  load_half(h2, t0 + B);
  float2 *out = reinterpret_cast<float2 *>(b.xspec) + (int64_t)(st.xspec_base - b.xspec_block0 + jblock) * B;
#pragma unroll
  for (int i = 0; i < E; ++i) {   // h1 <- s = w1 + w2, h2 <- d = w1 - w2
    const float2 p = h1[i], q = h2[i];
    h1[i] = make_float2(p.x + q.x, p.y + q.y);
    h2[i] = make_float2(p.x - q.x, p.y - q.y);
  }
  {
    float2 u[E];
    split_odd_input<G>(h2, u, s, w8, tid);
    fft_regs_to_regs<G, -1>(u, s, tw, tid);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      out[M + tid + T * i] = u[i];
    }
  }
  fft_regs_to_regs<G, -1>(h1, s, tw, tid);
  real_unpack_store_regs<G, L>(h1, s, tw.w0, tid, out);
}

template <int LOG2M>
__global__ __launch_bounds__((FftGeom<LOG2M - 1, 16>::T), AL_SPLIT_WAVES) void k_signal_spectra_split(al_batch b) {
  __shared__ float2 s[FftGeom<LOG2M - 1, 16>::LDS_ELEMS];
  signal_spectra_split_body<LOG2M>(b, s, blockIdx.x, blockIdx.y);
}

// Both forward transforms in ONE launch (they are independent): the 1 536 signal windows of a cfg2 scene are a 0.05 ms
// kernel of their own otherwise, too short to fill the chip.  Workgroup ids [0, n_sig) are signal jobs, the rest IR jobs.
template <int LOG2M>
__global__ __launch_bounds__((FftGeom<LOG2M - 1, 16>::T), AL_SPLIT_WAVES) void k_forward_spectra_split(al_batch b, int n_sig) {
  __shared__ float2 s[FftGeom<LOG2M - 1, 16>::LDS_ELEMS];
  __shared__ float red[48];
  const int id = blockIdx.x;
  if (id < n_sig) {
    signal_spectra_split_body<LOG2M>(b, s, id % b.max_nj, id / b.max_nj);
  } else {
    // Workgroup ids are dealt round-robin over the 8 XCDs; with the partition index as the plain remainder of the id an XCD would
    // only ever see partitions of one residue class (12 partitions per row: p mod 4 fixed per XCD), and the trimmed late
    // partitions of moving events (al_batch.emitter_parts) would be some XCDs' work only.  Rotating the index by the row gives every
    // XCD every partition: -2 % on cfg2's launch, -0.4 % on cfg3's (profiles/r04z_forward_id_mapping_ab.txt).
    const int q = id - n_sig, pc = b.n_partitions * b.n_capsules, row = q / b.n_partitions;
    ir_spectra_split_body<LOG2M>(b, s, red, (q + row) % b.n_partitions, row % b.n_capsules, q / pc);
  }
}

// ------------------------------------------------------------------ 5s. block synthesis, split layout
template <int LOG2M>
__global__ __launch_bounds__((FftGeom<LOG2M - 1, 16>::T), AL_SPLIT_WAVES) void k_block_synthesis_split(al_batch b) {
  using G = FftGeom<LOG2M - 1, 16>;
  using L = PlainSlots;
  constexpr int M = G::M, T = G::T, E = G::E, B = 2 * M;
  constexpr HalfTurnFactors<G> hf{};
  __shared__ float2 s[G::LDS_ELEMS];
  __shared__ float red[48];
  const int tid = threadIdx.x;
  const int k = blockIdx.x, c = blockIdx.y;
  const al_event ev = b.events[b.event0 + blockIdx.z];
  if (k >= ev.n_blocks) return;
  float *out = b.spatial + ev.out_off + (int64_t)c * ev.len;
  const int tbase = k * B;
  float asum = 0.f, amax = 0.f, bad = 0.f;
  if (ev.n_streams <= 0) {
    // no emitters: the clip is tiled over the capsules (synthesize.py:572-577)
    const float gain = b.streams[ev.stream0].gain * (b.clip_scale ? b.clip_scale[b.event0 + blockIdx.z] : 1.f);
    const float *a = b.audio + ev.audio_off;
    for (int i = tid; i < B; i += T) {
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
    const float2 *table = reinterpret_cast<const float2 *>(b.twiddle);
    FftTwiddles<G> tw;
    load_fft_twiddles<G, 1, 2>(tw, table, tid);
    const float2 w8 = table[tid];
    const float2 *y = reinterpret_cast<const float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + (int64_t)c * ev.n_blocks + k) * B;
    const float scale = 0.5f / (float)M;            // 1/(B/2) of each half-size inverse, and the 1/2 of (s - d)/2
    float2 u[E];
#pragma unroll
    for (int i = 0; i < E; ++i) u[i] = L::load_odd(y + M, tid + T * i);   // odd half, requested first
    float2 v[E];
    real_pack_load_regs<G, L>(y, v, s, tw.w0, tid, scale);
    fft_regs_to_regs<G, 1>(v, s, tw, tid);          // v[i] = (s[2m], s[2m+1]) / 2, m = tid + T*i
#pragma unroll
    for (int i = 0; i < E; ++i) u[i] = make_float2(u[i].x * scale, u[i].y * scale);
    fft_regs_to_regs<G, 1>(u, s, tw, tid);          // u[n'] / 2
    float *dl = reinterpret_cast<float *>(s);
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int n = tid + T * i;
      const float2 t = cmul(u[i], cconj(cmul(w8, make_float2(hf.c[i], -hf.s[i]))));   // * exp(+i pi n'/B)
      dl[n] = t.x;                                  // d[n'] / 2
      dl[n + M] = -t.y;                             // d[n' + B/2] / 2
    }
    __syncthreads();
    float2 pairs[2 * E];                            // synth_store_block reads the upper half of a register image
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float2 d = reinterpret_cast<const float2 *>(dl)[tid + T * i];     // (d[2m], d[2m+1]) / 2
      pairs[E + i] = make_float2(v[i].x - d.x, v[i].y - d.y);
    }
    const bool pair_ok = (((ev.out_off + (int64_t)c * ev.len) & 1) == 0);
    synth_store_block<SplitOut<G>>(pairs, out, ev, tbase, pair_ok, tid, asum, amax);
    bad = isfinite(asum) ? 0.f : 1.f;
    if (bad != 0.f) { asum = 0.f; amax = 0.f; }
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

}  // namespace al
