// Quad-tile transforms at B = 16384 (included by al_transforms.hip).
//
// Why: the accumulate prefers few, long partitions -- for IRs of 2.8-4.1 s (17..24 partitions at B = 8192, e.g. cfg5) only
// B = 16384 brings it back into the capsule loop's register tile (P <= 12, the signal window in registers across capsules:
// 5.2 instead of 4.1 TB/s on cfg5's accumulate).  But at B = 16384 the transforms of rounds 1-3 fall off their line: one
// 16384-point transform per window needs 139 KB of LDS (one workgroup per CU), two 8192-point ones (split layout) 68 KB each
// (two); cfg5 at B = 16384 through them: 17.0 ms per scene against 15.0 at B = 8192.  The 4096-point transform (35 KB of LDS,
// three passes of radix 16) is the one this chip runs well.  So a window's spectrum at B = 16384 is made of FOUR independent
// 4096-point transforms ("quad tiles"), the NATIVE layout of a block at this size (no slot map):
//   slots [0, Q)   T0 = W[4i]   = rFFT_{2Q}(a)[i],  a[t] = s[t] + s[t + 2Q]          (slot 0 packs W[0] and W[B])
//   slots [Q, 2Q)  T1 = W[8i+2] = FFT_Q( ((s0 - s2) - i (s1 - s3)) w^2 )
//   slots [2Q,3Q)  T2 = W[8i+1] = FFT_Q( ((d0 - i d2) + c8 (d1 - i d3)) w   )
//   slots [3Q,4Q)  T3 = W[8i+5] = FFT_Q( ((d0 - i d2) - c8 (d1 - i d3)) w^5 )
// with Q = B/4 = 4096, the window w = [w1, w2] of 2B samples, s = w1 + w2, d = w1 - w2, x_k = x[n + k Q], n < Q,
// w = e^{-i pi n / B}, c8 = e^{-i pi / 4} (an IR partition has w2 = 0: s = d = h).  Inverse: a from T0; (dd[n], dd[n+Q]) from T1
// with dd[t] = s[t] - s[t + 2Q]; v[n], v[n + Q] from T2 +- T3 with v[m] = (d[m] - i d[m + 2Q]) w_m; then the alias-free half
// w2 = (s - d) / 2.  (Checked in numpy before it was written; every row against the oracle in tests/test_gpu_mac_regimes.py.)
// The accumulate is element-wise on slots and only treats slot 0 specially: unchanged.
//
// Shape of the kernels (what was measured on the way: profiles/r04s_quad16_ab_v*.txt, profiles/HISTORY.md section 5.3):
//  - a window's samples are read ONCE and the inputs of all four transforms held in registers (128), the inverse keeps 64 partial
//    sums across its four transforms: two waves per SIMD (256 registers), not the three of the 8192-point split kernels;
//  - what the lower occupancy would expose is covered inside the workgroup: every barrier of the transforms waits for LDS traffic
//    only (block_barrier), so stores and the requests for the NEXT tile / IR partition stay in flight across them;
//  - both kernels use about a third of the chip's VALU issue slots (SQ_INSTS_VALU x 2 cycles per wave64 instruction on a SIMD-32,
//    profiles/pmc_traffic.json) and move their bytes at 5.1-5.3 TB/s: memory-side kernels, like the 8192-point split kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "al_common.h"
#include "al_fft.h"

#ifndef AL_Q16_WAVES
#define AL_Q16_WAVES 2   /* minimum waves per SIMD these kernels are compiled for: 64-128 values live across their transforms */
#endif

namespace al {

struct Quad16 {
  using G = FftGeom<12, 16>;                       // Q = 4096 points: 256 threads x 16 values, 35 KB of LDS
  static constexpr int Q = 4096, B = 16384, T = 256;
};

// Twists e^{-i pi K n / B}, K = 1, 2, 4, 5, at a thread's slots n = tid + 256 i: the thread's own power (e^{-i pi tid / B})^K, made once per
// window from ONE table entry, times the compile-time step e^{-i pi K i / 64}
constexpr double ct_cos_turn(int num, int den) {       // cos(pi num / den), any num >= 0
  num %= 2 * den;
  if (num > den) num = 2 * den - num;
  return 2 * num <= den ? ct_cospi(num, den) : -ct_cospi(den - num, den);
}
constexpr double ct_sin_turn(int num, int den) {       // sin(pi num / den), any num >= 0
  num %= 2 * den;
  const double sign = num > den ? -1.0 : 1.0;
  if (num > den) num = 2 * den - num;
  return sign * (2 * num <= den ? ct_sinpi(num, den) : ct_sinpi(den - num, den));
}
template <int K>
struct Quad16Steps {
  float c[16], s[16];
  constexpr Quad16Steps() : c{}, s{} {
    for (int i = 0; i < 16; ++i) {
      c[i] = (float)ct_cos_turn(K * i, 64);
      s[i] = (float)ct_sin_turn(K * i, 64);
    }
  }
};
template <int K>
__device__ __forceinline__ float2 quad16_twist(float2 wtk, int i) {
  constexpr Quad16Steps<K> ff{};
  return cmul(wtk, make_float2(ff.c[i], -ff.s[i]));
}
struct Quad16Twists {            // (e^{-i pi tid / B})^K for K = 1, 2, 5
  float2 w1, w2, w5;
  __device__ __forceinline__ explicit Quad16Twists(float2 wt) : w1(wt), w2(cmul(wt, wt)), w5(cmul(cmul(w2, w2), wt)) {}
};

// Transform inputs of tiles 1..3 at slot n from the four samples x_k = x[n + k Q] of s (tile 1) or d (tiles 2, 3); `w` = the twist of
// the tile at n: e^{-i pi n / B} to the power 2 (tile 1), 1 (tile 2), 5 (tile 3)
template <int TILE>
__device__ __forceinline__ float2 quad16_fold(float x0, float x1, float x2, float x3, float2 w) {
  constexpr float R2 = 0.70710678118654752440f;
  if constexpr (TILE == 1) return cmul(make_float2(x0 - x2, x3 - x1), w);
  const float2 p = make_float2(x0, -x2), q = make_float2(R2 * (x1 - x3), -R2 * (x1 + x3));
  if constexpr (TILE == 2) return cmul(cadd(p, q), w);
  return cmul(csub(p, q), w);
}

// One transform of a kernel that runs several: the pass factors are made opaque first, or the fifteen powers per pass derived from
// them would be computed ONCE for all the transforms of the kernel and held in registers across it (90 registers).
// LDS_ONLY: the passes' barriers wait for LDS traffic only (block_barrier), so global loads requested before the transform stay in flight.
template <int DIR, bool LDS_ONLY = false>
__device__ __forceinline__ void quad16_fft(float2 (&v)[16], float2 *s, FftTwiddles<Quad16::G> &tw, int tid) {
  tw.hide_from_hoisting();
  FftPasses<Quad16::G, DIR, 0, true, LDS_ONLY>::run(v, s, tw, tid);
}

// ------------------------------------------------------------------ forward: the four tiles of one window
// sk[k][i] = s[n + k Q], dk[k][i] = d[n + k Q] at n = tid + 256 i (an IR partition: the same array twice).  The samples are read
// ONCE (a workgroup that comes back for them a transform later finds most of them gone from the L2: +20 % on the kernel,
// profiles/r04s_quad16_ab_v1_reread.txt) and held across the transforms -- 64 values for an IR partition, 96 for a signal window --
// which is why these kernels are compiled for two waves per SIMD.  `prefetch(0)` is called once tiles 2 and 3 are stored and the
// samples are no longer needed, `prefetch(1)` after tile 1: the caller requests one half of its next window at each, into the
// registers that have become free.
template <bool SAME, class Prefetch>
__device__ __forceinline__ void quad16_forward_tiles(const float (&sk)[4][16], const float (&dk)[4][16], float2 *__restrict__ out, float2 *s,
                                                     FftTwiddles<Quad16::G> &tw, float2 wt, int tid, Prefetch &&prefetch) {
  using G = Quad16::G;
  constexpr int Q = Quad16::Q, T = Quad16::T;
  float2 z1[16], z2[16], z3[16];
  float a_lo[16], a_hi[16];          // a[n], a[n + Q] of a[t] = s[t] + s[t + 2Q]
  make_opaque(wt);                   // the twists are made here, per window
  const Quad16Twists tk(wt);
  // SAME (an IR partition, s = d = h): each tile's input is made from the 64 samples right before its transform, so that no more
  // than those 64 are held across a transform; else (a signal window: 64 + 64 samples) all of them now, 96 held across the first.
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    z2[i] = quad16_fold<2>(dk[0][i], dk[1][i], dk[2][i], dk[3][i], quad16_twist<1>(tk.w1, i));
    if constexpr (!SAME) {
      z3[i] = quad16_fold<3>(dk[0][i], dk[1][i], dk[2][i], dk[3][i], quad16_twist<5>(tk.w5, i));
      z1[i] = quad16_fold<1>(sk[0][i], sk[1][i], sk[2][i], sk[3][i], quad16_twist<2>(tk.w2, i));
      a_lo[i] = sk[0][i] + sk[2][i];
      a_hi[i] = sk[1][i] + sk[3][i];
    }
  }
  // every barrier below waits for LDS traffic only (block_barrier): stores and the caller's requests stay in flight across them
  quad16_fft<-1, true>(z2, s, tw, tid);
#pragma unroll
  for (int i = 0; i < 16; ++i) stream_store<4>(out + 2 * Q + tid + T * i, z2[i]);
  if constexpr (SAME) {
    pipeline_fence();
#pragma unroll
    for (int i = 0; i < 16; ++i) z3[i] = quad16_fold<3>(dk[0][i], dk[1][i], dk[2][i], dk[3][i], quad16_twist<5>(tk.w5, i));
  }
  quad16_fft<-1, true>(z3, s, tw, tid);
#pragma unroll
  for (int i = 0; i < 16; ++i) stream_store<4>(out + 3 * Q + tid + T * i, z3[i]);
  if constexpr (SAME) {
    pipeline_fence();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      z1[i] = quad16_fold<1>(sk[0][i], sk[1][i], sk[2][i], sk[3][i], quad16_twist<2>(tk.w2, i));
      a_lo[i] = sk[0][i] + sk[2][i];
      a_hi[i] = sk[1][i] + sk[3][i];
    }
  }
  prefetch(std::integral_constant<int, 0>{});
  pipeline_fence();
  quad16_fft<-1, true>(z1, s, tw, tid);
#pragma unroll
  for (int i = 0; i < 16; ++i) stream_store<4>(out + Q + tid + T * i, z1[i]);
  prefetch(std::integral_constant<int, 1>{});
  pipeline_fence();
  // a into the pair layout (a[2m], a[2m+1]), m = tid + 256 i, through the idle image (2Q floats): tile 0 is its real transform
  float *dl = reinterpret_cast<float *>(s);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    dl[tid + T * i] = a_lo[i];
    dl[tid + T * i + Q] = a_hi[i];
  }
  block_barrier<true>();
  float2 z0[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) z0[i] = reinterpret_cast<const float2 *>(dl)[tid + T * i];
  block_barrier<true>();
  quad16_fft<-1, true>(z0, s, tw, tid);
  real_unpack_store_regs<G, PlainSlots, true>(z0, s, tw.w0, tid, out);        // T0: slots [0, Q)
  block_barrier<true>();             // the image's last reads are over
}

// ------------------------------------------------------------------ 1q. IR partition spectra
// A workgroup walks a RUN of consecutive partitions of one IR row (the next one requested while this one is transformed).  The run
// index is the fastest-varying part of the workgroup id and ids are dealt round-robin over the 8 XCDs, so runs of unequal length put
// the long ones on some XCDs and the short ones on the others: 5 partitions as 4 + 1 send every run of four to four of the eight
// XCDs (4.8 ms on a 16 384-row batch; as one run of 5: 3.7 ms, as five of 1: 3.8 ms, and 3.7 ms again as 4 + 1 with the run index
// SLOWEST: profiles/r04z_forward_run_variants_moving.txt, r04z_unequal_runs_xcd_probe.txt).  So a row is cut into EQUAL runs of 2..6
// partitions where its partition count allows (cfg5: 12 as 6 + 6), the whole row otherwise -- shorter only if the batch has too few
// rows to fill the chip.
// AL_FLAG_IR_RUN(n) is the caller's own choice of run length.
static inline int quad16_run_len(const al_batch *b) {
  const int P = b->n_partitions, forced = (b->flags >> 24) & 0x7f;
  if (forced) return forced < P ? forced : P;
  const int64_t rows = (int64_t)b->n_capsules * b->n_emitters;
  int pick = 0;
  for (int n_runs = 1; n_runs <= P; ++n_runs) {
    if (P % n_runs || (rows * n_runs < 1024 && n_runs < P)) continue;   // equal runs only; enough workgroups to fill the chip
    const int run = P / n_runs;
    if (!pick) pick = run;                                               // the whole row, if nothing shorter divides it
    if (run >= 2 && run <= 6) return run;                                // long enough to hide the hand-over, short enough to spread
  }
  return pick ? pick : 1;
}
static inline int quad16_runs(const al_batch *b, int run_len) { return (b->n_partitions + run_len - 1) / run_len; }

__device__ __forceinline__ void ir_spectra_quad16_body(const al_batch &b, float2 *s, float *red, int run, int run_len, int c, int nz) {
  using G = Quad16::G;
  constexpr int Q = Quad16::Q, B = Quad16::B, T = Quad16::T;
  int tid = threadIdx.x;
  const int n_ir = b.emitter0 + nz;
  const float2 *table = reinterpret_cast<const float2 *>(b.twiddle);
  FftTwiddles<G> tw;
  load_fft_twiddles<G, -1, 4>(tw, table, tid);
  const float2 wt = table[tid];
  const float *row = b.ir + (int64_t)c * b.ir_stride_c + (int64_t)n_ir * b.ir_stride_n;
  const int p0 = run * run_len, p1 = min(p0 + run_len, b.n_partitions);
  const int p_live = (b.emitter_parts && b.n_partitions <= AL_SPARSE_MAX_PARTITIONS) ? b.emitter_parts[n_ir] : b.n_partitions;   // al_batch.emitter_parts
  // h[k][i] = h[p B + n + k Q], n = tid + 256 i, zeros past the IR's end; `half` 0: i < 8, 1: the rest
  auto request = [&](int p, float (&h)[4][16], int half) {
    const float *ir = row + (int64_t)p * B;
    const int remaining = b.ir_len - p * B;
    if (remaining >= B) {                              // workgroup-uniform
#pragma unroll
      for (int i = 8 * half; i < 8 * half + 8; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k][i] = ir[tid + T * i + Q * k];
    } else {
      const int last = remaining - 1;
#pragma unroll
      for (int i = 8 * half; i < 8 * half + 8; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = tid + T * i + Q * k;
          const float v = ir[min(t, last)];
          h[k][i] = t <= last ? v : 0.f;
        }
    }
  };
  float h[4][16];
  request(p0, h, 0);
  request(p0, h, 1);
#pragma unroll 1
  for (int p = p0; p < p1; ++p) {
    tid = opaque_lane(tid);                            // or every lane address of the body is hoisted out of the loop
    float energy = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) energy = fmaf(h[k][i], h[k][i], energy);
    {
      float mx = 0.f, zz = 0.f;
      block_reduce3(energy, mx, zz, red, tid, T);
      if (tid == 0) b.ir_energy[((int64_t)n_ir * b.n_capsules + c) * b.n_partitions + p] = energy;
    }
    float hn[4][16];
    if (p < p_live) {                                  // workgroup-uniform; else: no kept block hears it, its energy still counts
      float2 *out = reinterpret_cast<float2 *>(b.hspec) + (((int64_t)nz * b.n_capsules + c) * b.n_partitions + p) * B;
      quad16_forward_tiles<true>(h, h, out, s, tw, wt, tid, [&](auto half_c) { if (p + 1 < p1) request(p + 1, hn, decltype(half_c)::value); });
    } else {
      if (p + 1 < p1) {
        request(p + 1, hn, 0);
        request(p + 1, hn, 1);
      }
      // a trimmed partition runs no transform, so nothing else separates thread 0's reads of `red` above from the other
      // waves' writes for partition p + 1 (two trimmed partitions in a row of one run)
#ifndef AL_TEST_REVERT_Q16_BARRIER   /* tests/shake.py builds ONE variant without this barrier: the race it closes (found by a reader in
                                        round 4, by no test) must make tests/test_gpu_shake.py fail, or that test proves nothing */
      block_barrier<true>();
#endif
    }
    if (p + 1 < p1) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k][i] = hn[k][i];
    }
  }
}

// ------------------------------------------------------------------ 3q. signal block spectra
// A clip with its stream's gain and cross-fade envelope: the sample at absolute time t, 0 outside the clip.
struct Quad16Signal {
  const float *a, *w;
  int len, w_len, hop;
  float gain;
  bool moving;
  __device__ __forceinline__ float at(int t) const {
    const float x = a[min(max(t, 0), len - 1)];
    float g = (t >= 0 && t < len) ? gain : 0.f;
    if (moving) g *= stream_envelope(w, w_len, hop, max(t, 0));
    return x * g;
  }
};

// Any window but the interior ones of a static event (clip edges, cross-fade envelopes): every tile evaluates its own samples,
// one slot per trip of a ROLLED loop (unrolled, the 128 guarded, enveloped samples of a tile are all in flight at once: 290
// spilled registers), parked in the idle transform image until the tile's sixteen slots are there.
template <int TILE>
__device__ __forceinline__ void signal_tile_quad16(const Quad16Signal &sig, int t0, const float2 *__restrict__ table, float2 *out,
                                                   float2 *s, FftTwiddles<Quad16::G> &tw, int tid) {
  constexpr int Q = Quad16::Q, B = Quad16::B, T = Quad16::T;
  using G = Quad16::G;
#pragma unroll 1
  for (int i = 0; i < 16; ++i) {
    float2 v;
    if constexpr (TILE == 0) {      // (a[2m], a[2m+1]), m = tid + 256 i, a[t] = s[t] + s[t + 2Q], s = w1 + w2
      const int t = t0 + 2 * (tid + T * i);
      v = make_float2((sig.at(t) + sig.at(t + B)) + (sig.at(t + 2 * Q) + sig.at(t + 2 * Q + B)),
                      (sig.at(t + 1) + sig.at(t + 1 + B)) + (sig.at(t + 1 + 2 * Q) + sig.at(t + 1 + 2 * Q + B)));
    } else {
      const int t = t0 + tid + T * i;
      float x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w1 = sig.at(t + Q * k), w2 = sig.at(t + Q * k + B);
        x[k] = TILE == 1 ? w1 + w2 : w1 - w2;
      }
      const float2 w = table[tid + T * i], w2 = cmul(w, w);      // e^{-i pi n / B} at n = tid + 256 i, straight from the table
      v = quad16_fold<TILE>(x[0], x[1], x[2], x[3], TILE == 1 ? w2 : TILE == 2 ? w : cmul(cmul(w2, w2), w));
    }
    s[tid + T * i] = v;
  }
  float2 z[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = s[tid + T * i];     // the thread's own values: no barrier before, one behind
  __syncthreads();
  quad16_fft<-1>(z, s, tw, tid);
  if constexpr (TILE == 0) {
    real_unpack_store_regs<G>(z, s, tw.w0, tid, out);
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) out[TILE * Q + tid + T * i] = z[i];
  }
}

__device__ __forceinline__ void signal_spectra_quad16_body(const al_batch &b, float2 *s, int jblock, int stream_index) {
  using G = Quad16::G;
  constexpr int Q = Quad16::Q, B = Quad16::B, T = Quad16::T;
  const int tid = threadIdx.x;
  const al_stream st = b.streams[b.stream0 + stream_index];
  if (jblock >= st.n_j) return;
  const al_event ev = b.events[st.event];
  const float2 *table = reinterpret_cast<const float2 *>(b.twiddle);
  FftTwiddles<G> tw;
  load_fft_twiddles<G, -1, 4>(tw, table, tid);
  const int t0 = (st.j_lo + jblock - 1) * B;     // window [(j-1)B, (j+1)B): w1 = first B samples, w2 = the next B
  const bool moving = st.w_off >= 0 && st.w_len > 0;
  const float gain = st.gain * (b.clip_scale ? b.clip_scale[st.event] : 1.f);
  float2 *out = reinterpret_cast<float2 *>(b.xspec) + (int64_t)(st.xspec_base - b.xspec_block0 + jblock) * B;
  if (!moving && t0 >= 0 && t0 + 2 * B <= ev.len) {    // interior window of a stream without an envelope (workgroup-uniform): plain loads
    const float *a = b.audio + ev.audio_off + t0;
    float sk[4][16], dk[4][16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w1 = gain * a[tid + T * i + Q * k], w2 = gain * a[tid + T * i + Q * k + B];
        sk[k][i] = w1 + w2;
        dk[k][i] = w1 - w2;
      }
    quad16_forward_tiles<false>(sk, dk, out, s, tw, table[tid], tid, [](auto) {});
  } else {
    const Quad16Signal sig{b.audio + ev.audio_off, b.wtab + (moving ? st.w_off : 0), ev.len, st.w_len, b.hop, gain, moving};
    signal_tile_quad16<0>(sig, t0, table, out, s, tw, tid);
    signal_tile_quad16<1>(sig, t0, table, out, s, tw, tid);
    signal_tile_quad16<2>(sig, t0, table, out, s, tw, tid);
    signal_tile_quad16<3>(sig, t0, table, out, s, tw, tid);
  }
}

// both forward transforms in one launch, as k_forward_spectra_split: workgroups [0, n_sig) are signal windows, the rest runs of IR partitions
__global__ __launch_bounds__(256, AL_Q16_WAVES) void k_forward_spectra_quad16(al_batch b, int n_sig, int n_runs, int run_len) {
  __shared__ float2 s[Quad16::G::LDS_ELEMS];
  __shared__ float red[48];
  const int id = blockIdx.x;
  if (id < n_sig) {
    signal_spectra_quad16_body(b, s, id % b.max_nj, id / b.max_nj);
  } else {
    const int q = id - n_sig, rc = n_runs * b.n_capsules;
    ir_spectra_quad16_body(b, s, red, q % n_runs, run_len, (q / n_runs) % b.n_capsules, q / rc);
  }
}

__global__ __launch_bounds__(256, AL_Q16_WAVES) void k_ir_spectra_quad16(al_batch b, int run_len) {
  __shared__ float2 s[Quad16::G::LDS_ELEMS];
  __shared__ float red[48];
  ir_spectra_quad16_body(b, s, red, blockIdx.x, run_len, blockIdx.y, blockIdx.z);
}

__global__ __launch_bounds__(256, AL_Q16_WAVES) void k_signal_spectra_quad16(al_batch b) {
  __shared__ float2 s[Quad16::G::LDS_ELEMS];
  signal_spectra_quad16_body(b, s, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------ 5q. block synthesis
__global__ __launch_bounds__(256, AL_Q16_WAVES) void k_block_synthesis_quad16(al_batch b) {
  using G = Quad16::G;
  constexpr int Q = Quad16::Q, B = Quad16::B, T = Quad16::T;
  constexpr float R2 = 0.70710678118654752440f;
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
    load_fft_twiddles<G, 1, 4>(tw, table, tid);
    const float2 wt = table[tid];
    const float2 *y = reinterpret_cast<const float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + (int64_t)c * ev.n_blocks + k) * B;
    const float inv = 1.0f / (float)Q;                       // every Q-point inverse transform is unnormalised
    // x[n + k Q] = (s_k - d_k) / 2, accumulated tile by tile: acc[k][i] for n = tid + 256 i.  Every tile's slots are requested one
    // transform ahead of their use (the transforms' barriers wait for LDS traffic only, so the requests stay in flight).
    float acc[4][16];
    float2 z1[16];
    {   // tiles 2, 3: v[n], v[n + Q] = (z2 +- z3 w^-4) / 2;  d[m] - i d[m + 2Q] = v[m] conj(w_m)
      float2 z3[16], z2[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) z3[i] = stream_load<2>(y + 3 * Q + tid + T * i);
#pragma unroll
      for (int i = 0; i < 16; ++i) z2[i] = stream_load<2>(y + 2 * Q + tid + T * i);
      pipeline_fence();
      quad16_fft<1, true>(z3, s, tw, tid);
#pragma unroll
      for (int i = 0; i < 16; ++i) z1[i] = stream_load<2>(y + Q + tid + T * i);
      pipeline_fence();
      quad16_fft<1, true>(z2, s, tw, tid);
      float2 wl = wt;
      make_opaque(wl);          // the twists are made HERE, not held across the transforms
      const float2 wl2 = cmul(wl, wl), wl4 = cmul(wl2, wl2);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float2 cw = cconj(quad16_twist<1>(wl, i)), cw4 = cconj(quad16_twist<4>(wl4, i));
        const float2 a3 = cmul(z3[i], cw4);
        const float2 v0 = make_float2(z2[i].x + a3.x, z2[i].y + a3.y), v1 = make_float2(z2[i].x - a3.x, z2[i].y - a3.y);   // 2 v[n], 2 v[n+Q]
        const float2 p = cmul(v0, cw);                                           // 2 (d0 - i d2)
        const float2 q = cmul(cmul(v1, cw), make_float2(R2, R2));               // 2 (d1 - i d3): conj(w_{n+Q}) = conj(w_n) e^{+i pi/4}
        const float h = -0.25f * inv;                                           // - d_k / 2 with the factor 2 above
        acc[0][i] = h * p.x; acc[2][i] = -h * p.y; acc[1][i] = h * q.x; acc[3][i] = -h * q.y;
      }
    }
    float2 yk[G::H], ym[G::H];
    real_pack_issue<G, PlainSlots>(y, yk, ym, tid);                            // tile 0's slots, for after tile 1
    pipeline_fence();
    {   // tile 1: dd[n] - i dd[n + Q] = z1 conj(w^2);  s[t] = (a[t] + dd[t]) / 2, s[t + 2Q] = (a[t] - dd[t]) / 2
      quad16_fft<1, true>(z1, s, tw, tid);
      float2 wl = wt;
      make_opaque(wl);
      const float2 wl2 = cmul(wl, wl);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float2 u = cmul(z1[i], cconj(quad16_twist<2>(wl2, i)));          // dd[n] - i dd[n + Q]
        const float q4 = 0.25f * inv;                                          // s_k / 2 = (a +- dd) / 4
        acc[0][i] += q4 * u.x; acc[2][i] -= q4 * u.x;
        acc[1][i] -= q4 * u.y; acc[3][i] += q4 * u.y;
      }
    }
    {   // tile 0: a = irFFT_{2Q}(T0) as pairs (a[2m], a[2m+1]), m = tid + 256 i; into the strided layout through the (idle) image
      float2 z0[16];
      real_pack_finish<G>(yk, ym, z0, s, tw.w0, tid, 0.25f * inv);             // (a[2m], a[2m+1]) / 4 after the inverse passes
      quad16_fft<1, true>(z0, s, tw, tid);
      float *dl = reinterpret_cast<float *>(s);
#pragma unroll
      for (int i = 0; i < 16; ++i) reinterpret_cast<float2 *>(dl)[tid + T * i] = z0[i];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float lo = dl[tid + T * i], hi = dl[tid + T * i + Q];            // a[n] / 4, a[n + Q] / 4
        acc[0][i] += lo; acc[2][i] += lo;
        acc[1][i] += hi; acc[3][i] += hi;
      }
    }
    // The thread holds x[n + k Q], n = tid + 256 i: 4-byte stores, a wave's 64 consecutive samples each (whatever the row's alignment)
    if (tbase + B <= ev.valid_len) {   // interior block (workgroup-uniform)
      float *o = out + tbase + tid;
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float x = acc[kq][i];
          stream_store<16>(o + Q * kq + T * i, x);
          asum += fabsf(x);
          amax = fmaxf(amax, fabsf(x));
        }
      }
    } else {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = tbase + Q * kq + tid + T * i;
          const float x = t < ev.valid_len ? acc[kq][i] : 0.f;
          if (t < ev.len) {
            out[t] = x;
            asum += fabsf(x);
            amax = fmaxf(amax, fabsf(x));
          }
        }
      }
    }
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
