// HIP kernels + C ABI of the MI355X synthesis path (gfx950 only).  See DESIGN.md for the data
// layout and per-kernel rooflines, include/audiblelight_hip.h for the boundary.
//
// Pipeline per batch of events (uniformly partitioned overlap-save, block B = M = 2^LOG2M):
//   k_ir_spectra      H[n,c,p]  = rFFT_2B([h[pB:(p+1)B], 0])        + partial sum of h^2
//   k_emitter_gains   g[n]      = 1 / mean_c(||h_{n,c}|| + tiny)      (normalize_irs)
//   k_signal_spectra  X[s,j]    = rFFT_2B(gain * env_s * a[(j-1)B:(j+1)B])
//   k_spectral_mac    Y[e,c,k]  = sum_s g[n_s] sum_p X[s,k-p] * H[n_s,c,p]
//   k_block_synthesis x[e,c,kB:(k+1)B] = irFFT_2B(Y[e,c,k])[B:2B]    + partial |x| statistics
//   k_event_levels    scale[e]  = apply_snr o db_to_multiplier from sum|x|, max|x|
//   k_mixdown         scene[c,t] (+)= sum_e scale[e] * x[e,c,t-start_e]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>


#include "al_common.h"
#include "al_fft.h"
#include "al_bigfft.h"
#include "al_stft.h"

namespace al {

// ------------------------------------------------------------------ twiddle table
__global__ void k_twiddle_init(float2 *tw, int m) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) {
    double s, c;
    sincospi(-(double)k / (double)m, &s, &c);
    tw[k] = make_float2((float)c, (float)s);
  }
}

// A float64 scalar as the float32 the kernels multiply samples by, kept FINITE.  The reference forms these scalars in float64 --
// 1 / tiny = 4.5e307 for an all-zero IR or clip, 10^(dB/20) / tiny for a silent render -- and multiplies ZEROS by them: silence
// stays silence.  As float32 they would be +-inf and inf * 0 = NaN, so they saturate at +-FLT_MAX (FLT_MAX * 0 = 0).
__device__ __forceinline__ float finite_f32(double v) {
  if (v != v) return (float)v;     // NaN stays NaN: non-finite INPUT must still fail the finite check (librosa.util.valid_audio)
  return (float)fmin(fmax(v, -3.4028234663852886e38), 3.4028234663852886e38);
}

// ------------------------------------------------------------------ 2. emitter gains (normalize_irs)
// one wave per emitter: g = 1 / mean_c( sqrt(sum_t h^2) + tiny(float64) )   (synthesize.py:425-428)
// mode 0: g (single GPU); 1: emitter_gain[n] := sum over THIS rank's capsules of the norms (to be all-reduced);
// 2: emitter_gain[n] := total_capsules / emitter_gain[n] (the reduced sum), for capsule-sharded scenes
// An IR whose gain would not fit float32 (mean norm below 3e-39: every capsule's IR all zeros, or float32 denormals) gets gain 0:
// the reference divides its zeros by tiny and keeps zeros; a saturated gain would overflow the signal spectra it multiplies.
__device__ __forceinline__ float emitter_gain_of(double capsules, double norm_sum) {
  const double g = capsules / norm_sum;
  return (g <= 3.4028234663852886e38 || g != g) ? (float)g : 0.0f;   // NaN (a NaN in the IR) stays NaN and fails the finite check
}

__global__ __launch_bounds__(64) void k_emitter_gains(al_batch b, int mode, int total_capsules) {
  const int n = b.emitter0 + blockIdx.x, lane = threadIdx.x;
  if (mode == 2) {
    if (lane == 0)
      b.emitter_gain[n] = (b.flags & AL_FLAG_NO_IR_NORM) ? 1.0f : emitter_gain_of((double)total_capsules, (double)b.emitter_gain[n]);
    return;
  }
  double acc = 0.0;
  for (int c = lane; c < b.n_capsules; c += 64) {
    const float *e = b.ir_energy + ((int64_t)n * b.n_capsules + c) * b.n_partitions;
    double sum = 0.0;
    for (int p = 0; p < b.n_partitions; ++p) sum += (double)e[p];
    acc += sqrt(sum) + 2.2250738585072014e-308;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) {
    if (mode == 1) b.emitter_gain[n] = (float)acc;
    else b.emitter_gain[n] = (b.flags & AL_FLAG_NO_IR_NORM) ? 1.0f : emitter_gain_of((double)b.n_capsules, acc);
  }
}

// ------------------------------------------------------------------ 4. frequency-domain accumulate
// Y[k] = sum_p X[k-p] * H[p] is a Toeplitz product per frequency bin.  One thread owns one bin and
// walks (k-tile x p-tile) pairs: the KT accumulators and PT partition spectra of the pair stay in
// registers and the KT+PT-1 signal blocks on its anti-diagonals are loaded ONCE each, so a pair
// costs KT+2*PT-1 loads for KT*PT complex FMAs (static register indices throughout).
// Bin 0 packs (DC, Nyquist): two independent real products.
// VB = bins per thread (1: float2 accesses, 2: float4 accesses of two adjacent bins).
// a += x * h (complex) as TWO v_pk_fma_f32 whose operand halves are picked by op_sel / negated by neg_lo: no swizzled
// copies of x or h exist in registers (left to itself hipcc keeps (x.x, x.x) and (-x.y, x.y) for every resident spectrum,
// doubling its register cost: profiles/r01_mac_variants.txt).
__device__ __forceinline__ void cfma_packed(float2 &a, const float2 &x, const float2 &h) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f av = {a.x, a.y};
  const v2f xv = {x.x, x.y}, hv = {h.x, h.y};
  // lo: x.x*h.x + a.x          hi: x.x*h.y + a.y
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(av) : "v"(xv), "v"(hv));
  // lo: -x.y*h.y + a.x         hi: x.y*h.x + a.y
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(av) : "v"(xv), "v"(hv));
  a = make_float2(av.x, av.y);
#else
  cfma(a, x, h);
#endif
}

template <int VB> struct BinVec;
template <> struct BinVec<1> {
  float2 a;
  __device__ __forceinline__ static BinVec zero() { return BinVec{make_float2(0.f, 0.f)}; }
  __device__ __forceinline__ static BinVec load(const float2 *p) { return BinVec{*p}; }
  __device__ __forceinline__ void store(float2 *p) const { stream_store<1>(p, a); }
  __device__ __forceinline__ void scale(float g) { a.x *= g; a.y *= g; }
  // BIN0: this wave may hold bin 0 (packed DC/Nyquist: two real products); every other wave takes the plain path
  template <bool BIN0>
  __device__ __forceinline__ void fma(const BinVec &x, const BinVec &h, bool packed) {
    if (BIN0 && packed) { a.x = fmaf(x.a.x, h.a.x, a.x); a.y = fmaf(x.a.y, h.a.y, a.y); } else cfma(a, x.a, h.a);
  }
};
template <> struct BinVec<2> {
  float2 a, c;
  __device__ __forceinline__ static BinVec zero() { return BinVec{make_float2(0.f, 0.f), make_float2(0.f, 0.f)}; }
  __device__ __forceinline__ static BinVec load(const float2 *p) {
    const float4 v = *reinterpret_cast<const float4 *>(p);
    return BinVec{make_float2(v.x, v.y), make_float2(v.z, v.w)};
  }
  __device__ __forceinline__ void store(float2 *p) const { stream_store<1>(reinterpret_cast<float4 *>(p), make_float4(a.x, a.y, c.x, c.y)); }
  __device__ __forceinline__ void scale(float g) { a.x *= g; a.y *= g; c.x *= g; c.y *= g; }
  template <bool BIN0>
  __device__ __forceinline__ void fma(const BinVec &x, const BinVec &h, bool packed) {
    if (BIN0 && packed) { a.x = fmaf(x.a.x, h.a.x, a.x); a.y = fmaf(x.a.y, h.a.y, a.y); } else cfma(a, x.a, h.a);
    cfma(c, x.c, h.c);
  }
  template <bool BIN0>
  __device__ __forceinline__ void fma_packed(const BinVec &x, const BinVec &h, bool packed) {
    if (BIN0 && packed) { a.x = fmaf(x.a.x, h.a.x, a.x); a.y = fmaf(x.a.y, h.a.y, a.y); } else cfma_packed(a, x.a, h.a);
    cfma_packed(c, x.c, h.c);
  }
};

// k_spectral_mac_static takes the one-emitter events when the flag is set and the partitions fit one register tile
__host__ __device__ __forceinline__ bool static_mac_active(const al_batch &b) {
  // up to 21 partitions through the LDS-DMA kernel (it reads the rows past an odd partition count from the all-zero block),
  // up to 16 through the register-staged one when the caller gave no zero block.  22..24 stay on the tile kernels: three
  // units of 8 fit (ends of the 35-block window in LDS) but only tie them at C = 32 and lose 5 % on cfg5 (profiles/r03_p24_ab.txt)
  return (b.flags & AL_FLAG_STATIC_MAC) && b.n_partitions <= (b.hspec_zero_block >= 0 ? 21 : 16) && b.log2_block >= 9;
}

// KSPLIT: every k-tile is its own workgroup (blockIdx.y = c * n_ktiles + tile) instead of a loop inside the thread.
// Workgroups of one bin tile share blockIdx.x, hence (round-robin dispatch) an XCD and its L2, and the tiles of one
// (event, capsule) are adjacent in dispatch order: the second..n-th read of the partition spectra hits L2.
template <int KT, int PT, int VB, bool KSPLIT, bool BIN0>
__device__ __forceinline__ void spectral_mac_body(const al_batch &b) {
  using V = BinVec<VB>;
  const int M = 1 << b.log2_block;
  const int f = (blockIdx.x * 256 + threadIdx.x) * VB;
  const int n_ktiles = KSPLIT ? (b.max_blocks + KT - 1) / KT : 1;
  const int c = blockIdx.y / n_ktiles;
  const al_event ev = b.events[b.event0 + blockIdx.z];
  if (ev.n_streams <= 0) return;
  if (ev.n_streams > 1 && ev.reserved == 1 && b.n_partitions <= AL_SPARSE_MAX_PARTITIONS) return;  // k_spectral_mac_moving
  if (ev.n_streams == 1 && static_mac_active(b)) return;     // k_spectral_mac_static
  const float2 *__restrict__ X = reinterpret_cast<const float2 *>(b.xspec);
  const float2 *__restrict__ H = reinterpret_cast<const float2 *>(b.hspec);
  float2 *__restrict__ Y = reinterpret_cast<float2 *>(b.yspec);
  const int K = ev.n_blocks, P = b.n_partitions;
  const bool packed = (f == 0);  // bin 0 holds (DC, Nyquist): two independent real products
  const int k_first = KSPLIT ? (blockIdx.y % n_ktiles) * KT : 0;
  const int k_limit = KSPLIT ? min(K, k_first + KT) : K;

  for (int k0 = k_first; k0 < k_limit; k0 += KT) {
    V acc[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) acc[kk] = V::zero();
    for (int l = 0; l < ev.n_streams; ++l) {
      const al_stream st = b.streams[ev.stream0 + l];
      const int jlo = st.j_lo, jhi = st.j_lo + st.n_j;  // non-zero signal blocks [jlo, jhi)
      if (jhi <= jlo) continue;
      // partitions that can meet this k-tile: k0+kk-p in [jlo, jhi)
      const int plo = max(0, k0 - jhi + 1), phi = min(P - 1, k0 + KT - 1 - jlo);
      if (plo > phi) continue;
      const float g = b.emitter_gain[st.emitter];
      const float2 *hp = H + (((int64_t)(st.emitter - b.emitter0) * b.n_capsules + c) * P) * M + f;
      const float2 *xp = X + (int64_t)(st.xspec_base - b.xspec_block0 - jlo) * M + f;
      for (int p0 = plo; p0 <= phi; p0 += PT) {
        V h[PT];
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) {
          // unconditional load at a clamped partition, zeroed by the select: keeps all PT loads in flight
          h[pp] = V::load(hp + (int64_t)min(p0 + pp, phi) * M);
          h[pp].scale((p0 + pp <= phi) ? g : 0.f);
        }
        const int jbase = k0 - p0 - (PT - 1);  // signal block of anti-diagonal jj is jbase + jj
        // The KT+PT-1 signal blocks are fetched in groups of XG, one group ahead of the FMAs that
        // consume them (explicit double buffer): the loads are L2 hits with ~1 us latency under load,
        // and a wave that waits for them one by one is latency-bound, not bandwidth-bound.
        constexpr int XG = 8 / VB, NJ = KT + PT - 1, NG = (NJ + XG - 1) / XG;
        auto fetch = [&](int jj) -> V {
          const int j = jbase + jj;
          V x = V::load(xp + (int64_t)min(max(j, jlo), jhi - 1) * M);  // clamped, unconditional
          x.scale((j >= jlo && j < jhi) ? 1.f : 0.f);
          return x;
        };
        V xa[XG], xb[XG];
        static_for<XG>([&](auto i_c) {
          constexpr int i = decltype(i_c)::value;
          if constexpr (i < NJ) xa[i] = fetch(i);
        });
        static_for<NG>([&](auto g_c) {
          constexpr int g_ = decltype(g_c)::value;
          static_for<XG>([&](auto i_c) {  // prefetch group g+1
            constexpr int i = decltype(i_c)::value;
            if constexpr ((g_ + 1) * XG + i < NJ) xb[i] = fetch((g_ + 1) * XG + i);
          });
          static_for<XG>([&](auto i_c) {  // consume group g
            constexpr int i = decltype(i_c)::value;
            constexpr int jj = g_ * XG + i;
            if constexpr (jj < NJ) {
              static_for<KT>([&](auto kk_c) {
                constexpr int kk = decltype(kk_c)::value;
                constexpr int pp = kk + (PT - 1) - jj;
                if constexpr (pp >= 0 && pp < PT) acc[kk].template fma<BIN0>(xa[i], h[pp], packed);
              });
            }
          });
          static_for<XG>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            if constexpr ((g_ + 1) * XG + i < NJ) xa[i] = xb[i];
          });
        });
      }
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
      if (k0 + kk < K) acc[kk].store(Y + ((int64_t)(ev.yspec_base - b.yspec_block0) + (int64_t)c * K + k0 + kk) * M + f);
  }
}

// Bin 0 needs two real products instead of a complex one.  Only the first wave of the first bin tile can hold it:
// that wave runs the BIN0 instantiation (per-lane select), every other wave the plain complex path -- folding the
// select into the common path costs two extra FMAs and two v_cndmask per product for EVERY bin (measured: 43 % of
// the kernel's vector instructions).
template <int KT, int PT, int VB, bool KSPLIT = false>
__global__ __launch_bounds__(256) void k_spectral_mac(al_batch b) {
  if (blockIdx.x == 0 && threadIdx.x < 64) spectral_mac_body<KT, PT, VB, KSPLIT, true>(b);
  else spectral_mac_body<KT, PT, VB, KSPLIT, false>(b);
}

// ------------------------------------------------------------------ 4a. accumulate for static events, capsule loop
// A static event has ONE stream, so the signal blocks a (k-tile, bin tile) needs -- the KT+PT-1 blocks on its anti-
// diagonals -- are the same for every capsule.  One workgroup therefore owns (event, k-tile, bin tile) and LOOPS over the
// capsules with that window held in registers (loaded once, already multiplied by the emitter gain and zeroed where
// k - p leaves the clip: no per-capsule masks or gain multiplies).  What this buys over one workgroup per capsule
// (profiles/r02_mac.txt): the signal spectra leave L2 once instead of C times, the workgroup start-up (three dependent
// table reads) and the dispatch of 32x as many workgroups disappear, and because nothing waits on X any more the
// partition spectrum h[p] of the NEXT capsule is requested the moment the last product with h[p] of this one has been
// issued -- the H stream, the FMAs and the Y stores of consecutive capsules overlap inside one wave.
// PT is the batch's partition count itself (one instantiation per P = 1..12): every h[pp] is a real partition, nothing in the
// loop is masked.  (A 12-wide tile with the missing partitions zeroed by a multiply, selected from a zero block or skipped by
// a uniform branch spilled 104-192 B per lane and ran 25-70 % slower than the tile kernels at P = 10, profiles/r02_mac.txt.)
// NKTW: k-tiles per workgroup (256 threads each).  Two k-tiles of one (event, bin tile) read the SAME partition spectra;
// in one workgroup, kept in step by a barrier per capsule, the second read of every line is an L1 hit on the same CU
// instead of a second trip to L2 / HBM by another workgroup that may have drifted away.
// The capsule-loop kernels' bin tile: blockIdx.x rotated by blockIdx.z.  Workgroup ids are dealt round-robin over the 8 XCDs and the
// grids are 16 (or 32) bin tiles wide, so with the plain index an XCD would only ever touch two of the sixteen 4 KB columns of every
// spectrum block; rotated, every XCD sees every column: -3 % on the accumulate of cfg2, cfg4 and cfg5
// (profiles/r04z_rotated_ids_mac_synth_ab.txt).  (The tile kernel k_spectral_mac keeps the plain index: it WANTS the k-tiles of one
// bin tile on one XCD, for the L2 hits on H; the sliding-window kernel is 3.5 % slower rotated, r04z_rotated_ids_moving_ab.txt.)
__device__ __forceinline__ int rotated_bin_tile() { return (int)((blockIdx.x + blockIdx.z) % gridDim.x); }

template <int KT, int PT, bool BIN0, int NKTW>
__device__ __forceinline__ void spectral_mac_static_body(const al_batch &b, int bx) {
  using V = BinVec<2>;
  constexpr int NJ = KT + PT - 1;
  const int M = 1 << b.log2_block;
  const int lane256 = threadIdx.x & 255, sub = threadIdx.x >> 8;
  const int f = (bx * 256 + lane256) * 2;
  const int n_cs = gridDim.z / b.n_events;                      // capsule ranges per event (small batches)
  const int e = blockIdx.z / n_cs, cs = blockIdx.z % n_cs;
  const al_event ev = b.events[b.event0 + e];
  if (ev.n_streams != 1) return;                                // moving events: k_spectral_mac / k_spectral_mac_moving
  const int K = ev.n_blocks, P = b.n_partitions, C = b.n_capsules;
  const int k0 = (blockIdx.y * NKTW + sub) * KT;
  if (NKTW == 1 && k0 >= K) return;
  const bool active = k0 < K;                                   // NKTW > 1: an idle half still joins the barriers
  const int c_begin = (int)((int64_t)cs * C / n_cs), c_end = (int)((int64_t)(cs + 1) * C / n_cs);
  const al_stream st = b.streams[ev.stream0];
  const int jlo = st.j_lo, jhi = st.j_lo + st.n_j;
  const int plo = max(0, k0 - jhi + 1), phi = min(P - 1, k0 + KT - 1 - jlo);
  const bool packed = (f == 0);
  const float2 *__restrict__ X = reinterpret_cast<const float2 *>(b.xspec) + (int64_t)(st.xspec_base - b.xspec_block0 - jlo) * M + f;
  const float2 *__restrict__ H = reinterpret_cast<const float2 *>(b.hspec) + ((int64_t)(st.emitter - b.emitter0) * C * P) * M + f;
  float2 *__restrict__ Y = reinterpret_cast<float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + k0) * M + f;
  const float g = b.emitter_gain[st.emitter];
  const bool single = phi - plo < PT;                           // one partition tile: the window survives the capsule loop
  // (plo > phi cannot happen for a static event, whose signal blocks are [0, K): plo = 0 <= phi)

  V xw[NJ];
  auto load_window = [&](int p0) {                              // xw[jj] = g * X[k0 - p0 - (PT-1) + jj], 0 outside the clip
    const int jbase = k0 - p0 - (PT - 1);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int j = jbase + jj;
      xw[jj] = V::load(X + (int64_t)min(max(j, jlo), jhi - 1) * M);
      xw[jj].scale((j >= jlo && j < jhi) ? g : 0.f);
    }
  };
  auto h_load = [&](int c, int p) { return V::load(H + ((int64_t)c * P + min(p, P - 1)) * M); };
  V h[PT];
  if (single && active) {
    load_window(plo);
#pragma unroll
    for (int pp = 0; pp < PT; ++pp) h[pp] = h_load(c_begin, plo + pp);
  }
  for (int c = c_begin; c < c_end; ++c) {
    if (NKTW > 1) __syncthreads();                              // both k-tiles start the capsule together
    if (!active) continue;
    const int cn = min(c + 1, c_end - 1);                       // capsule whose spectra are requested during this one
    V acc[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) acc[kk] = V::zero();
    for (int p0 = plo; p0 <= phi; p0 += PT) {
      if (!single) {
        load_window(p0);
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) h[pp] = h_load(c, p0 + pp);
      }
      static_for<PT>([&](auto pp_c) {
        constexpr int pp = decltype(pp_c)::value;
        static_for<KT>([&](auto kk_c) {
          constexpr int kk = decltype(kk_c)::value;
          acc[kk].template fma_packed<BIN0>(xw[kk + (PT - 1) - pp], h[pp], packed);   // X[k0 + kk - (p0 + pp)]
        });
        if (single) h[pp] = h_load(cn, plo + pp);                // h[pp] is free: fetch the next capsule's
      });
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
      if (k0 + kk < K) acc[kk].store(Y + ((int64_t)c * K + kk) * M);
  }
}

template <int KT, int PT, int NKTW>
__global__ __launch_bounds__(256 * NKTW, 2) void k_spectral_mac_static(al_batch b) {
  const int bx = rotated_bin_tile();
  if (bx == 0 && (threadIdx.x & 255) < 64) spectral_mac_static_body<KT, PT, true, NKTW>(b, bx);
  else spectral_mac_static_body<KT, PT, false, NKTW>(b, bx);
}

// Variant of the two-k-tile workgroup for clips of more than 24 blocks (several workgroups per (event, bin tile), all reading
// the same partition spectra): the 512 threads copy the (PT x 512 slot) tile of capsule c+2 into a ring of three LDS stages
// while capsule c is multiplied, so H enters the CU once instead of twice (the second k-tile's L1 hit) and no partition
// spectrum waits in registers.  Equal to the register version at K <= 24, 8-10 % faster beyond (profiles/r02_mac.txt 10).
template <int KT, int PT, int UNITS, bool BIN0>
__device__ __forceinline__ void spectral_mac_static_lds_body(const al_batch &b, float4 *hbuf, int bx) {
  // UNITS = 2: 13..16 partitions as two units of PT = ceil(P / 2) per capsule (the pipeline step is a unit; for odd P the
  // last unit's missing partition is stored as zeros in its LDS stage, so nothing in the products is masked)
  using V = BinVec<2>;
  constexpr int PALL = UNITS * PT, NJ = KT + PALL - 1, STAGE = PT * 256, PER = (STAGE + 511) / 512;
  const int M = 1 << b.log2_block;
  const int lane256 = threadIdx.x & 255, sub = threadIdx.x >> 8;
  const int f = (bx * 256 + lane256) * 2;
  const int n_cs = gridDim.z / b.n_events;
  const int e = blockIdx.z / n_cs, cs = blockIdx.z % n_cs;
  const al_event ev = b.events[b.event0 + e];
  if (ev.n_streams != 1) return;
  const int K = ev.n_blocks, P = b.n_partitions, C = b.n_capsules;
  const int k0 = (blockIdx.y * 2 + sub) * KT;
  const bool active = k0 < K;
  const int c_begin = (int)((int64_t)cs * C / n_cs), c_end = (int)((int64_t)(cs + 1) * C / n_cs);
  const al_stream st = b.streams[ev.stream0];
  const int jlo = st.j_lo, jhi = st.j_lo + st.n_j;
  const bool packed = (f == 0);
  const float2 *__restrict__ X = reinterpret_cast<const float2 *>(b.xspec) + (int64_t)(st.xspec_base - b.xspec_block0 - jlo) * M + f;
  const float2 *__restrict__ Htile = reinterpret_cast<const float2 *>(b.hspec) + ((int64_t)(st.emitter - b.emitter0) * C * P) * M + bx * 512;
  float2 *__restrict__ Y = reinterpret_cast<float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + k0) * M + f;
  const float g = b.emitter_gain[st.emitter];
  V xw[NJ];
  if (active) {
    const int jbase = k0 - (PALL - 1);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int j = jbase + jj;
      xw[jj] = V::load(X + (int64_t)min(max(j, jlo), jhi - 1) * M);
      xw[jj].scale((j >= jlo && j < jhi) ? g : 0.f);
    }
  }
  static_assert(PER <= 6, "staging registers are named, not indexed (an indexed array stayed in scratch memory)");
  float4 g0, g1, g2, g3, g4, g5;
  g0 = g1 = g2 = g3 = g4 = g5 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n_units = UNITS * (c_end - c_begin);
  // unit n = (capsule c_begin + n / UNITS, partitions [(n % UNITS) * PT, +PT)); row r of the copy is partition p0 + r
#define AL_FETCH1(R, I, C_, P0_)                                                                                             \
  if ((I) < PER && (STAGE % 512 == 0 || (int)threadIdx.x + 512 * (I) < STAGE)) {                                             \
    const int q_ = (int)threadIdx.x + 512 * (I), p_ = (P0_) + (q_ >> 8);                                                     \
    R = (UNITS == 1 || p_ < P) ? *reinterpret_cast<const float4 *>(Htile + ((int64_t)(C_) * P + min(p_, P - 1)) * M + (q_ & 255) * 2) \
                               : make_float4(0.f, 0.f, 0.f, 0.f);                                                            \
  }
#define AL_FETCH(UNIT)                                                                                                       \
  { const int n_ = min((UNIT), n_units - 1), cc_ = c_begin + n_ / UNITS, p0_ = (n_ % UNITS) * PT;                            \
    AL_FETCH1(g0, 0, cc_, p0_) AL_FETCH1(g1, 1, cc_, p0_) AL_FETCH1(g2, 2, cc_, p0_) AL_FETCH1(g3, 3, cc_, p0_)               \
    AL_FETCH1(g4, 4, cc_, p0_) AL_FETCH1(g5, 5, cc_, p0_) }
#define AL_STASH1(R, I, S)                                                                                                   \
  if ((I) < PER && (STAGE % 512 == 0 || (int)threadIdx.x + 512 * (I) < STAGE)) hbuf[(S) * STAGE + threadIdx.x + 512 * (I)] = R;
#define AL_STASH(STAGE_INDEX)                                                                                                \
  { const int ss_ = (STAGE_INDEX); AL_STASH1(g0, 0, ss_) AL_STASH1(g1, 1, ss_) AL_STASH1(g2, 2, ss_) AL_STASH1(g3, 3, ss_)   \
    AL_STASH1(g4, 4, ss_) AL_STASH1(g5, 5, ss_) }
  AL_FETCH(0)
  AL_STASH(0)
  AL_FETCH(1)
  AL_STASH(1)
  int n = 0;                                                    // unit being multiplied
  for (int c = c_begin; c < c_end; ++c) {
    V acc[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) acc[kk] = V::zero();
    static_for<UNITS>([&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      const int cur = n % 3, nxt = (n + 2) % 3;
      AL_FETCH(n + 2)                                           // in flight during this unit's products (past the end the last
                                                                // unit is fetched again: unconditional code)
      __syncthreads();                                          // stage `cur` is complete, stage `nxt` is no longer read
      if (active) {
        const float4 *hs = hbuf + cur * STAGE + lane256;
        float4 hv = hs[0], hn = hv;
        static_for<PT>([&](auto pp_c) {
          constexpr int pp = decltype(pp_c)::value;
          if constexpr (pp + 1 < PT) {                          // the next partition's LDS read is issued before this one's
            hn = hs[(pp + 1) * 256];                            // products, not a few instructions before its first use
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
          const V h{make_float2(hv.x, hv.y), make_float2(hv.z, hv.w)};
          static_for<KT>([&](auto kk_c) {
            constexpr int kk = decltype(kk_c)::value;
            acc[kk].template fma_packed<BIN0>(xw[kk + (PALL - 1) - (u * PT + pp)], h, packed);
          });
          hv = hn;
        });
      }
      AL_STASH(nxt)
      ++n;
    });
    if (active) {
#pragma unroll
      for (int kk = 0; kk < KT; ++kk)
        if (k0 + kk < K) acc[kk].store(Y + ((int64_t)c * K + kk) * M);
    }
  }
#undef AL_FETCH
#undef AL_STASH
#undef AL_FETCH1
#undef AL_STASH1
}

template <int KT, int PT, int UNITS = 1>
__global__ __launch_bounds__(512, 2) void k_spectral_mac_static_lds(al_batch b) {
  __shared__ float4 hbuf[3 * PT * 256];
  const int bx = rotated_bin_tile();
  if (bx == 0 && (threadIdx.x & 255) < 64) spectral_mac_static_lds_body<KT, PT, UNITS, true>(b, hbuf, bx);
  else spectral_mac_static_lds_body<KT, PT, UNITS, false>(b, hbuf, bx);
}

// ------------------------------------------------------------------ 4a'. capsule loop fed by LDS-DMA
// The same loop with the partition spectra brought into the LDS ring by LDS-DMA (global_load_lds_dwordx4: the data never
// passes through a VGPR and the instruction returns at once).  A wave has only two register sets' worth of room, so the
// register versions request capsule c+1's spectra WHILE capsule c is multiplied -- one iteration (a few microseconds) of
// flight time, less than the latency of a loaded HBM (profiles/r02_mac.txt 9).  Here the request for unit n+2 is issued at the
// start of unit n and retired by a COUNTED s_waitcnt at the start of unit n+2: two iterations in flight, no staging
// registers, no ds_write pass.  Protocol per unit n (cdna_hip_programming.md section 5, "Pipelining across barriers"):
//     s_waitcnt vmcnt(N)   this wave's DMA pieces of unit n have landed (N = the VMEM operations it issued after them:
//                          the pieces of unit n+1 and the Y stores in between; VMEM operations of a wave retire in order)
//     s_barrier            ... and so have every other wave's; everybody is done reading the stage unit n+2 will overwrite
//     issue DMA of unit n+2 -> stage (n+2) % 3;   multiply unit n out of stage n % 3;   store Y at the end of a capsule
// The DMA is inline asm (hipcc would drain every outstanding one with vmcnt(0) before the first LDS read it knows to depend
// on it); the waits are therefore counted by hand.  Every wave issues the same number of pieces (the last piece is fetched
// again where PT * 4 is not a multiple of 8) and every half issues exactly its own number of stores per capsule.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_spectral_mac_static_glds counts stores in vmcnt and uses global_load_lds_dwordx4: gfx950 only (a target with a separate store counter would read stale LDS)"
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// 64 lanes x 16 B from (uniform base in SGPRs) + (lane offset in ONE VGPR shared by every piece) into LDS at lds_dst + lane * 16:
// all the address arithmetic of a piece is scalar
__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  al::shake(105);   // (test builds only, al_common.h: no memory operation, so the counted waits below are not disturbed)
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  al::shake(106);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#endif

// NL: the NL signal blocks at EACH end of the window live in LDS instead of registers.  Block jj of the window meets
// min(jj + 1, NJ - jj, KT) products per capsule, so the ends are the cheap ones to re-read: NL = 3 costs 12 extra
// ds_read_b128 per capsule and frees 24 VGPRs, which is what the 32-block window of 19..21 partitions (NL = 2) needs to stay
// out of scratch memory (a spill reload would also drain the LDS-DMA in flight: hipcc waits vmcnt(0) for it).
template <int KT, int PT, int UNITS, bool BIN0, bool ZERO_ROWS = (UNITS > 1), int NL = 0, int NKTW = 2>
__device__ __forceinline__ void spectral_mac_static_glds_body(const al_batch &b, float4 *hbuf, float4 *xbuf, int bx) {
  using V = BinVec<2>;
  constexpr int NWAVES = 4 * NKTW;      // NKTW k-tiles of 256 threads per workgroup
  constexpr int PALL = UNITS * PT, NJ = KT + PALL - 1, STAGE = PT * 256, PIECES = PT * 4, PER_WAVE = (PIECES + NWAVES - 1) / NWAVES;
  const int M = 1 << b.log2_block;
  const int lane256 = threadIdx.x & 255, sub = threadIdx.x >> 8, lane = threadIdx.x & 63;
  const int f = (bx * 256 + lane256) * 2;
  const int n_cs = gridDim.z / b.n_events;
  const int e = blockIdx.z / n_cs, cs = blockIdx.z % n_cs;
  const al_event ev = b.events[b.event0 + e];
  if (ev.n_streams != 1) return;
  const int K = ev.n_blocks, P = b.n_partitions, C = b.n_capsules;
  const int k0 = (blockIdx.y * NKTW + sub) * KT;
  const bool active = k0 < K;
  const int n_stores = active ? min(KT, K - k0) : 0;            // Y stores this half issues per capsule (workgroup-half uniform)
  const int c_begin = (int)((int64_t)cs * C / n_cs), c_end = (int)((int64_t)(cs + 1) * C / n_cs);
  const al_stream st = b.streams[ev.stream0];
  const int jlo = st.j_lo, jhi = st.j_lo + st.n_j;
  const bool packed = (f == 0);
  const float2 *__restrict__ X = reinterpret_cast<const float2 *>(b.xspec) + (int64_t)(st.xspec_base - b.xspec_block0 - jlo) * M + f;
  const float2 *__restrict__ Htile = reinterpret_cast<const float2 *>(b.hspec) + ((int64_t)(st.emitter - b.emitter0) * C * P) * M + bx * 512;
  const float2 *__restrict__ Hzero = reinterpret_cast<const float2 *>(b.hspec) + (int64_t)max(b.hspec_zero_block, 0) * M;   // rows past P (odd P in units)
  float2 *__restrict__ Y = reinterpret_cast<float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + k0) * M + f;
  const float g = b.emitter_gain[st.emitter];
  V xw[NJ - 2 * NL];                                            // window blocks [NL, NJ - NL); the ends are in xbuf
  if (active) {
    const int jbase = k0 - (PALL - 1);
    static_for<NJ>([&](auto jj_c) {
      constexpr int jj = decltype(jj_c)::value;
      const int j = jbase + jj;
      V x = V::load(X + (int64_t)min(max(j, jlo), jhi - 1) * M);
      x.scale((j >= jlo && j < jhi) ? g : 0.f);
      if constexpr (jj < NL) xbuf[jj * (256 * NKTW) + threadIdx.x] = make_float4(x.a.x, x.a.y, x.c.x, x.c.y);
      else if constexpr (jj >= NJ - NL) xbuf[(jj - (NJ - 2 * NL)) * (256 * NKTW) + threadIdx.x] = make_float4(x.a.x, x.a.y, x.c.x, x.c.y);
      else xw[jj - NL] = x;
    });
  }
  auto window = [&](auto jj_c) -> V {                          // a thread reads back only what it wrote: no barrier needed
    constexpr int jj = decltype(jj_c)::value;
    if constexpr (jj < NL || jj >= NJ - NL) {
      const float4 v = xbuf[(jj < NL ? jj : jj - (NJ - 2 * NL)) * (256 * NKTW) + threadIdx.x];
      return V{make_float2(v.x, v.y), make_float2(v.z, v.w)};
    } else {
      return xw[jj - NL];
    }
  };
  const int n_units = UNITS * (c_end - c_begin);
#if defined(__HIP_DEVICE_COMPILE__)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __attribute__((address_space(3))) float4 *hbuf_lds = (__attribute__((address_space(3))) float4 *)hbuf;
  const unsigned lds_base = (unsigned)(uintptr_t)hbuf_lds;      // byte offset of the ring inside the workgroup's LDS
#else
  const int wave = (int)(threadIdx.x >> 6);
#endif
  // unit n = (capsule c_begin + n / UNITS, partitions [(n % UNITS) * PT, +PT)); piece q of row r = 1 KB = 64 lanes x 16 B
  auto issue = [&](int n) {
    const int n_ = min(n, n_units - 1), cc = c_begin + n_ / UNITS, p0 = (n_ % UNITS) * PT, stage = n % 3;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int piece = min(wave + NWAVES * i, PIECES - 1), r = piece >> 2, q = piece & 3, p = p0 + r;
      const float2 *src = ((!ZERO_ROWS || p < P) ? Htile + ((int64_t)cc * P + p) * M : Hzero) + q * 128;   // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
      glds16(src, (unsigned)lane * 16u, lds_base + (unsigned)(stage * STAGE + r * 256 + q * 64) * 16u);
#else
      hbuf[stage * STAGE + r * 256 + q * 64 + lane] = *reinterpret_cast<const float4 *>(src + lane * 2);
#endif
    }
  };
  // N of the counted wait in front of unit n: VMEM operations issued after the pieces of unit n, i.e. the pieces of unit
  // n+1 and the stores that fell between them; exact for UNITS == 1 (a capsule per unit: the stores of the two previous
  // capsules), the pieces alone otherwise (stricter than needed when a capsule ended in between: still correct)
  auto wait_unit = [&](int unit) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (UNITS == 1) {
      switch (min(unit, 2) * n_stores) {     // the first two capsules have fewer stores behind them
#define AL_WAIT_CASE(S) case S: wait_vm<PER_WAVE + S>(); break;
        AL_WAIT_CASE(0) AL_WAIT_CASE(1) AL_WAIT_CASE(2) AL_WAIT_CASE(3) AL_WAIT_CASE(4) AL_WAIT_CASE(5) AL_WAIT_CASE(6)
        AL_WAIT_CASE(7) AL_WAIT_CASE(8) AL_WAIT_CASE(9) AL_WAIT_CASE(10) AL_WAIT_CASE(11) AL_WAIT_CASE(12) AL_WAIT_CASE(14)
        AL_WAIT_CASE(16) AL_WAIT_CASE(18) AL_WAIT_CASE(20) AL_WAIT_CASE(22) AL_WAIT_CASE(24)
#undef AL_WAIT_CASE
        default: wait_vm<PER_WAVE>(); break;
      }
    } else {
      wait_vm<PER_WAVE>();
    }
    __builtin_amdgcn_s_barrier();
#else
    __syncthreads();
#endif
  };
  static_assert(KT <= 12, "the counted waits enumerate at most 12 stores per capsule");
  static_assert(PER_WAVE + 24 <= 63, "vmcnt is a 6-bit counter: the pieces of a unit plus two capsules' stores must fit it");
  issue(0);
  issue(1);
  int n = 0;
  for (int c = c_begin; c < c_end; ++c) {
    V acc[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) acc[kk] = V::zero();
    static_for<UNITS>([&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      wait_unit(n);                                             // unit n is in stage n % 3; stage (n + 2) % 3 is no longer read
      issue(n + 2);
      if (active) {
        const float4 *hs = hbuf + (n % 3) * STAGE + lane256;
        float4 hv = hs[0], hn = hv;
        static_for<PT>([&](auto pp_c) {
          constexpr int pp = decltype(pp_c)::value;
          if constexpr (pp + 1 < PT) {
            hn = hs[(pp + 1) * 256];
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
          const V h{make_float2(hv.x, hv.y), make_float2(hv.z, hv.w)};
          static_for<KT>([&](auto kk_c) {
            constexpr int kk = decltype(kk_c)::value;
            acc[kk].template fma_packed<BIN0>(window(std::integral_constant<int, kk + (PALL - 1) - (u * PT + pp)>{}), h, packed);
          });
          hv = hn;
        });
      }
      ++n;
    });
    if (active) {
#pragma unroll
      for (int kk = 0; kk < KT; ++kk)
        if (kk < n_stores) acc[kk].store(Y + ((int64_t)c * K + kk) * M);
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  wait_vm<0>();   // the two re-fetched units past the end must have landed before the LDS goes back to the next workgroup
#endif
}

// ZERO_ROWS: the partition count is not a multiple of UNITS * PT, the missing rows of the last unit come from the all-zero block
template <int KT, int PT, int UNITS = 1, bool ZERO_ROWS = (UNITS > 1), int NL = 0, int NKTW = 2>
__global__ __launch_bounds__(256 * NKTW, NKTW) void k_spectral_mac_static_glds(al_batch b) {
  __shared__ float4 hbuf[3 * PT * 256];
  __shared__ float4 xbuf[NL > 0 ? 2 * NL * 256 * NKTW : 1];
  const int bx = rotated_bin_tile();
  if (bx == 0 && (threadIdx.x & 255) < 64) spectral_mac_static_glds_body<KT, PT, UNITS, true, ZERO_ROWS, NL, NKTW>(b, hbuf, xbuf, bx);
  else spectral_mac_static_glds_body<KT, PT, UNITS, false, ZERO_ROWS, NL, NKTW>(b, hbuf, xbuf, bx);
}

// ------------------------------------------------------------------ 4b. accumulate for moving events
// A moving event is N streams (one per IR) whose clips are only a few blocks long (the cross-fade window of
// that IR) and whose first blocks j_lo are non-decreasing.  One thread owns one bin (pair) of one capsule and
// walks the streams in order with a SLIDING window of W = NJW + PT - 1 output accumulators anchored at the
// current stream's j_lo: blocks that fall behind the window are complete and are written out once.  Every H,
// X and Y value moves exactly once and every register index is static.
template <int NJW, int PT, int VB, bool BIN0>
__device__ __forceinline__ void spectral_mac_moving_body(const al_batch &b, int4 *tab, float *gains) {
  using V = BinVec<VB>;
  constexpr int W = NJW + PT - 1;
  const int M = 1 << b.log2_block;
  const int f = (blockIdx.x * 256 + threadIdx.x) * VB;
  const int c = blockIdx.y;
  const al_event ev = b.events[b.event0 + blockIdx.z];
  if (ev.n_streams <= 1 || ev.reserved != 1) return;  // static / dense events: k_spectral_mac
  const float2 *__restrict__ X = reinterpret_cast<const float2 *>(b.xspec);
  const float2 *__restrict__ H = reinterpret_cast<const float2 *>(b.hspec);
  float2 *__restrict__ Y = reinterpret_cast<float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + (int64_t)c * ev.n_blocks) * M + f;
  const int K = ev.n_blocks, P = b.n_partitions;
  const bool packed = (f == 0);
  V acc[W];
#pragma unroll
  for (int w = 0; w < W; ++w) acc[w] = V::zero();
  int kbase = 0;  // output block held in acc[0]
  for (int l0 = 0; l0 < ev.n_streams; l0 += 64) {
    __syncthreads();
    if (threadIdx.x < 64 && l0 + (int)threadIdx.x < ev.n_streams) {
      const al_stream st = b.streams[ev.stream0 + l0 + threadIdx.x];
      tab[threadIdx.x] = make_int4(st.j_lo, st.n_j, st.emitter - b.emitter0, st.xspec_base - b.xspec_block0);
      gains[threadIdx.x] = b.emitter_gain[st.emitter];
    }
    __syncthreads();
    const int nl = min(64, ev.n_streams - l0);
    for (int l = 0; l < nl; ++l) {
      const int4 t = tab[l];
      const int jlo = t.x, nj = t.y;
      if (nj <= 0) continue;
      // retire the blocks before this stream's first block
      while (kbase < jlo) {
        if (kbase < K) acc[0].store(Y + (int64_t)kbase * M);
#pragma unroll
        for (int w = 0; w + 1 < W; ++w) acc[w] = acc[w + 1];
        acc[W - 1] = V::zero();
        ++kbase;
      }
      // partitions of this IR that reach a block the event keeps (pad_or_truncate, synthesize.py:590, drops everything from
      // block K on): partition p of a stream that starts at block j_lo only feeds blocks >= j_lo + p.  The others are not read.
      const int pl = min(P, K - jlo);
      if (pl <= 0) continue;
      const float g = gains[l];
      const float2 *hp = H + (((int64_t)t.z * b.n_capsules + c) * P) * M + f;
      const float2 *xp = X + (int64_t)t.w * M + f;
      V h[PT], x[NJW];
#pragma unroll
      for (int pp = 0; pp < PT; ++pp) {
        h[pp] = V::load(hp + (int64_t)min(pp, pl - 1) * M);
        h[pp].scale(pp < pl ? g : 0.f);
      }
#pragma unroll
      for (int jj = 0; jj < NJW; ++jj) {
        x[jj] = V::load(xp + (int64_t)min(jj, nj - 1) * M);
        x[jj].scale(jj < nj ? 1.f : 0.f);
      }
#pragma unroll
      for (int jj = 0; jj < NJW; ++jj)
#pragma unroll
        for (int pp = 0; pp < PT; ++pp) acc[jj + pp].template fma<BIN0>(x[jj], h[pp], packed);
    }
  }
#pragma unroll
  for (int w = 0; w < W; ++w)
    if (kbase + w < K) acc[w].store(Y + (int64_t)(kbase + w) * M);
  // blocks beyond the last window (no stream reaches them) are zero
  for (int k = kbase + W; k < K; ++k) V::zero().store(Y + (int64_t)k * M);
}

template <int NJW, int PT, int VB>
__global__ __launch_bounds__(256) void k_spectral_mac_moving(al_batch b) {
  __shared__ int4 tab[64];    // {j_lo, n_j, emitter - emitter0, xspec_base - xspec_block0}
  __shared__ float gains[64];
  // both instantiations run the same barriers in the same order, so splitting the workgroup by wave is safe
  if (blockIdx.x == 0 && threadIdx.x < 64) spectral_mac_moving_body<NJW, PT, VB, true>(b, tab, gains);
  else spectral_mac_moving_body<NJW, PT, VB, false>(b, tab, gains);
}

// ------------------------------------------------------------------ 6. event levels
// Composite of apply_snr (synthesize.py:40-49) and db_to_multiplier (synthesize.py:52-68) as chained
// at synthesize.py:594-599, evaluated in float64 from the deterministic partial statistics.
// mode 0: reduce + law (single GPU); 1: reduce only; 2: law only, from event_stats, with `total_capsules` rows
__global__ __launch_bounds__(64) void k_event_levels(al_batch b, int mode, int total_capsules) {
  const int e = b.event0 + blockIdx.x;
  const al_event ev = b.events[e];
  const int lane = threadIdx.x;
  double *o = b.event_stats + 4 * (int64_t)e;
  double sum = 0.0, bad = 0.0;
  float mx = 0.f;
  if (mode != 2) {
    const int n = b.n_capsules * ev.n_blocks;
    const float *pp = b.partials + 4 * (int64_t)ev.part_base;
    for (int i = lane; i < n; i += 64) {
      sum += (double)pp[4 * i];
      mx = fmaxf(mx, pp[4 * i + 1]);
      bad += (double)pp[4 * i + 2];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sum += __shfl_down(sum, off, 64);
      mx = fmaxf(mx, __shfl_down(mx, off, 64));
      bad += __shfl_down(bad, off, 64);
    }
  }
  if (lane == 0) {
    if (mode == 2) {
      sum = o[0];
      mx = (float)o[1];
      bad = o[2];
    }
    if (mode == 1) {
      o[0] = sum;
      o[1] = (double)mx;
      o[2] = bad;
      o[3] = 0.0;
      return;
    }
    const double rows = mode == 2 ? (double)total_capsules : (double)b.n_capsules;
    const double snr = (double)ev.snr;
    const double peak = fmax((double)mx, 1e-15);
    const double s1 = snr / peak;                                   // apply_snr
    const double mean_abs = fabs(s1) * sum / (rows * (double)ev.len);
    const double s2 = pow(10.0, ((double)ev.ref_db + snr) / 20.0) / (mean_abs + 2.2250738585072014e-308);
    o[0] = sum;
    o[1] = (double)mx;
    o[2] = bad;
    o[3] = s2;
    // a silent render (sum|x| = 0: zero clip or zero IRs) has s2 = 10^(dB/20) / tiny: the reference multiplies its zeros by it
    b.event_scale[e] = finite_f32(s1 * s2);
  }
}

// ------------------------------------------------------------------ 7. mixdown
// One workgroup per (capsule, tile of m.tile = 4096 samples).  A thread owns 4 runs of 4 consecutive
// samples (16 accumulators); events that overlap the tile are walked in insertion order with their slot
// scalars in SGPRs, each adding scale * x with dword-aligned 16-byte loads (an event starts at an
// arbitrary sample, so its rows are not 16-byte aligned against the scene).
struct __attribute__((packed, aligned(4))) f4u {
  float x, y, z, w;
};

__global__ __launch_bounds__(256) void k_mixdown(al_mix m) {
  const int tile = blockIdx.x, c = blockIdx.y;
  const int lo = m.tile_ptr[tile], hi = m.tile_ptr[tile + 1];
  const int t_begin = tile * m.tile;
  float *row = m.scene + (int64_t)c * m.n_samples;
  constexpr int RUNS = 4;                       // m.tile == 4 * 256 * RUNS
  float4 acc[RUNS];
  const bool whole = (t_begin + m.tile <= m.n_samples) && ((m.n_samples & 3) == 0);  // workgroup-uniform
  const float amb_scale = m.ambience ? m.ambience_scale[c] : 0.f;   // per capsule: peak normalisation x noise-floor multiplier
  const float *amb = m.ambience ? m.ambience + (int64_t)c * m.n_samples : row;
#pragma unroll
  for (int r = 0; r < RUNS; ++r) {
    const int t = t_begin + 4 * (threadIdx.x + 256 * r);
    acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m.ambience) {   // the reference adds the ambience to the zeroed float32 buffer first (synthesize.py:335-356)
      if (whole) {
        const float4 nz = *reinterpret_cast<const float4 *>(amb + t);
        acc[r] = make_float4(amb_scale * nz.x, amb_scale * nz.y, amb_scale * nz.z, amb_scale * nz.w);
      } else {
        if (t < m.n_samples) acc[r].x = amb_scale * amb[t];
        if (t + 1 < m.n_samples) acc[r].y = amb_scale * amb[t + 1];
        if (t + 2 < m.n_samples) acc[r].z = amb_scale * amb[t + 2];
        if (t + 3 < m.n_samples) acc[r].w = amb_scale * amb[t + 3];
      }
    }
    if (m.accumulate) {
      if (whole) {
        acc[r] = *reinterpret_cast<const float4 *>(row + t);
      } else {
        if (t < m.n_samples) acc[r].x = row[t];
        if (t + 1 < m.n_samples) acc[r].y = row[t + 1];
        if (t + 2 < m.n_samples) acc[r].z = row[t + 2];
        if (t + 3 < m.n_samples) acc[r].w = row[t + 3];
      }
    }
  }
  for (int q = lo; q < hi; ++q) {
    const int sl = m.tile_events[q];
    if (c >= m.slot_rows[sl]) continue;
    const int start = m.slot_start[sl], count = m.slot_count[sl];
    const float scale = m.event_scale[m.slot_event[sl]];
    const float *x = m.spatial + m.slot_src[sl] + (int64_t)c * m.slot_len[sl];
#pragma unroll
    for (int r = 0; r < RUNS; ++r) {
      const int rel = t_begin + 4 * (threadIdx.x + 256 * r) - start;
      if (rel >= 0 && rel + 3 < count) {
        const f4u v = *reinterpret_cast<const f4u *>(x + rel);
        acc[r].x = fmaf(scale, v.x, acc[r].x);
        acc[r].y = fmaf(scale, v.y, acc[r].y);
        acc[r].z = fmaf(scale, v.z, acc[r].z);
        acc[r].w = fmaf(scale, v.w, acc[r].w);
      } else if (rel > -4 && rel < count) {  // run straddles the start or the end of the slot
        if (rel >= 0 && rel < count) acc[r].x = fmaf(scale, x[rel], acc[r].x);
        if (rel + 1 >= 0 && rel + 1 < count) acc[r].y = fmaf(scale, x[rel + 1], acc[r].y);
        if (rel + 2 >= 0 && rel + 2 < count) acc[r].z = fmaf(scale, x[rel + 2], acc[r].z);
        if (rel + 3 >= 0 && rel + 3 < count) acc[r].w = fmaf(scale, x[rel + 3], acc[r].w);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RUNS; ++r) {
    const int t = t_begin + 4 * (threadIdx.x + 256 * r);
    if (whole) {
      stream_store<64>(reinterpret_cast<float4 *>(row + t), acc[r]);
    } else {
      if (t < m.n_samples) row[t] = acc[r].x;
      if (t + 1 < m.n_samples) row[t + 1] = acc[r].y;
      if (t + 2 < m.n_samples) row[t + 2] = acc[r].z;
      if (t + 3 < m.n_samples) row[t + 3] = acc[r].w;
    }
  }
}

// ------------------------------------------------------------------ helpers
__global__ __launch_bounds__(256) void k_scale(float *x, int64_t n, const float *scale) {
  const float s = *scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= s;
}

__global__ __launch_bounds__(256) void k_scale_d(float *x, int64_t n, const double *scale) {
  const float s = finite_f32(*scale);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= s;
}

__global__ __launch_bounds__(256) void k_axpy(float *y, const float *x, const float *a, int64_t n) {
  const float s = *a;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = fmaf(s, x[i], y[i]);
}

// y[r, :] += a[r] * x[r, :]: an ambience with its per-channel scale (second and further ambiences of a scene)
__global__ __launch_bounds__(256) void k_axpy_rows(float *y, const float *x, const float *a, int64_t cols) {
  const float s = a[blockIdx.y];
  const int64_t base = (int64_t)blockIdx.y * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cols; i += (int64_t)gridDim.x * 256)
    y[base + i] = fmaf(s, x[base + i], y[base + i]);
}

// Per-channel multiplier of an ambience from its row statistics {sum|x|, max|x|, ...} (al_row_stats), one wave, float64:
// the per-channel peak normalisation ch / max(|ch| + tiny) (ambience.py:211-214) and db_to_multiplier(ref_db, mean|normalised|)
// (synthesize.py:350-356) as ONE scalar per channel, so the noise is neither rescaled in place nor read by the host.
__global__ __launch_bounds__(64) void k_ambience_scales(const double *__restrict__ stats, int rows, int64_t cols, float ref_db,
                                                        int normalize, float *__restrict__ scales) {
  const int lane = threadIdx.x;
  double acc = 0.0;
  for (int c = lane; c < rows; c += 64) {
    const double inv = normalize ? 1.0 / (stats[4 * c + 1] + 2.2250738585072014e-308) : 1.0;
    acc += stats[4 * c] * inv;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double total;
  if (lane == 0) total = acc;
  __syncthreads();
  const double mean_abs = total / ((double)rows * (double)cols);
  const double mult = pow(10.0, (double)ref_db / 20.0) / (mean_abs + 2.2250738585072014e-308);
  for (int c = lane; c < rows; c += 64) {
    const double inv = normalize ? 1.0 / (stats[4 * c + 1] + 2.2250738585072014e-308) : 1.0;
    scales[c] = finite_f32(normalize == 2 ? inv : mult * inv);   // 2: the peak normalisation alone; a silent channel stays silent
  }
}

constexpr int ROW_CHUNK = 16384;  // samples per partial of k_row_stats

__global__ __launch_bounds__(256) void k_row_stats(const float *x, int64_t cols, float *partials) {
  __shared__ float red[48];
  const int nchunks = (int)((cols + ROW_CHUNK - 1) / ROW_CHUNK);
  const int chunk = blockIdx.x % nchunks, r = blockIdx.x / nchunks;  // rows may exceed the 65535 limit of grid.y
  const int64_t lo = (int64_t)chunk * ROW_CHUNK, hi = lo + ROW_CHUNK < cols ? lo + ROW_CHUNK : cols;
  const float *row = x + (int64_t)r * cols;
  float asum = 0.f, amax = 0.f, bad = 0.f, sq = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = row[i];
    asum += fabsf(v);
    amax = fmaxf(amax, fabsf(v));
    bad += isfinite(v) ? 0.f : 1.f;
    sq = fmaf(v, v, sq);
  }
  block_reduce3(asum, amax, bad, red, threadIdx.x, 256);
  __syncthreads();
  float z0 = 0.f, z1 = 0.f;
  block_reduce3(sq, z0, z1, red, threadIdx.x, 256);
  if (threadIdx.x == 0) {
    float *pp = partials + 4 * ((int64_t)r * nchunks + chunk);
    pp[0] = asum;
    pp[1] = amax;
    pp[2] = bad;
    pp[3] = sq;
  }
}

__global__ __launch_bounds__(64) void k_row_stats_final(const float *partials, int nchunks, double *out) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float *pp = partials + 4 * (int64_t)r * nchunks;
  double sum = 0.0, bad = 0.0, sq = 0.0;
  float mx = 0.f;
  for (int i = lane; i < nchunks; i += 64) {
    sum += (double)pp[4 * i];
    mx = fmaxf(mx, pp[4 * i + 1]);
    bad += (double)pp[4 * i + 2];
    sq += (double)pp[4 * i + 3];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
    mx = fmaxf(mx, __shfl_down(mx, off, 64));
    bad += __shfl_down(bad, off, 64);
    sq += __shfl_down(sq, off, 64);
  }
  if (lane == 0) {
    out[4 * r + 0] = sum;
    out[4 * r + 1] = (double)mx;
    out[4 * r + 2] = bad;
    out[4 * r + 3] = sq;
  }
}

// ------------------------------------------------------------------ sample-wise clip operations (A13/A14)
__device__ __forceinline__ float fade_in_curve(int shape, float r) {  // augmentation.py:1490-1508
  switch (shape) {
    case AL_FADE_EXPONENTIAL: return exp2f(r - 1.f) * r;
    case AL_FADE_LOGARITHMIC: return log10f(0.1f + r) + 1.f;
    case AL_FADE_QUARTER_SINE: return sinpif(0.5f * r);
    case AL_FADE_HALF_SINE: return 0.5f * sinpif(r - 0.5f) + 0.5f;
    default: return r;
  }
}
__device__ __forceinline__ float fade_out_curve(int shape, float r) {  // augmentation.py:1510-1528
  switch (shape) {
    case AL_FADE_EXPONENTIAL: return exp2f(-r) * (1.f - r);
    case AL_FADE_LOGARITHMIC: return log10f(1.1f - r) + 1.f;
    case AL_FADE_QUARTER_SINE: return sinpif(0.5f * r + 0.5f);
    case AL_FADE_HALF_SINE: return 0.5f * sinpif(r + 0.5f) + 0.5f;
    default: return 1.f - r;
  }
}

struct FxArgs {
  int op;
  float p0;
  int n_in, n_out, shape_in, shape_out;
};

__global__ __launch_bounds__(256) void k_fx_pointwise(const float *src, float *dst, int64_t n, FxArgs a) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    float x = src[a.op == AL_FX_REVERSE ? n - 1 - t : t];
    switch (a.op) {
      case AL_FX_GAIN: x *= a.p0; break;
      case AL_FX_INVERT: x = -x; break;
      case AL_FX_CLIP: x = fminf(fmaxf(x, -a.p0), a.p0); break;
      case AL_FX_TANH: x = tanhf(a.p0 * x); break;
      case AL_FX_BITCRUSH: x = rintf(x * a.p0) / a.p0; break;
      case AL_FX_FADE: {
        float g = 1.f;
        if (a.n_in > 0 && a.shape_in != AL_FADE_NONE && t < a.n_in) {
          const float r = a.n_in > 1 ? (float)t / (float)(a.n_in - 1) : 0.f;  // np.linspace(0, 1, n_in)
          g *= fminf(fmaxf(fade_in_curve(a.shape_in, r), 0.f), 1.f);
        }
        if (a.n_out > 0 && a.shape_out != AL_FADE_NONE && t >= n - a.n_out) {
          const float r = a.n_out > 1 ? (float)(t - (n - a.n_out)) / (float)(a.n_out - 1) : 0.f;
          g *= fminf(fmaxf(fade_out_curve(a.shape_out, r), 0.f), 1.f);
        }
        x *= g;
      } break;
      case AL_FX_PREEMPH: {
        if (t == 0) x = x + (2.f * x - (n > 1 ? src[1] : x));
        else x = fmaf(-a.p0, src[t - 1], x);
      } break;
      default: break;
    }
    dst[t] = x;
  }
}

// y[n] = x[n] + c*y[n-1] minus the extrapolation correction: one workgroup, each thread owns a
// contiguous run; carries are chained by thread 0 (1024 runs), then folded back in.
__global__ __launch_bounds__(1024) void k_fx_deemph(const float *src, float *dst, int64_t n, float c) {
  __shared__ float tail[1024], decay[1024], carry[1024];
  const int tid = threadIdx.x;
  const int64_t run = (n + 1023) / 1024;
  const int64_t lo = (int64_t)tid * run, hi = lo + run < n ? lo + run : n;
  float y = 0.f, d = 1.f;
  for (int64_t t = lo; t < hi; ++t) {
    y = fmaf(c, y, src[t]);
    d *= c;
    dst[t] = y;
  }
  tail[tid] = y;
  decay[tid] = d;
  __syncthreads();
  if (tid == 0) {
    float acc = 0.f;
    for (int i = 0; i < 1024; ++i) {
      carry[i] = acc;  // state entering run i
      acc = fmaf(decay[i], acc, tail[i]);
    }
  }
  __syncthreads();
  const float x0 = src[0], x1 = n > 1 ? src[1] : src[0];
  const float corr = ((2.f - c) * x0 - x1) / (3.f - c);
  float pw = c;                      // c^(t - lo + 1)
  float cn = powf(c, (float)lo);     // c^t
  const float cin = carry[tid];
  for (int64_t t = lo; t < hi; ++t) {
    dst[t] = fmaf(cin, pw, dst[t]) - corr * cn;
    pw *= c;
    cn *= c;
  }
}

__global__ __launch_bounds__(256) void k_frame_shuffle(const float *src, float *dst, int64_t n, int frame_len,
                                                       int row_len, const int32_t *rows, int n_rows) {
  const int64_t total = (int64_t)n_rows * row_len;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int64_t u = t % total;
    const int q = (int)(u / row_len), j = (int)(u - (int64_t)q * row_len);
    const int r = rows[2 * q], mode = rows[2 * q + 1];
    const int jj = mode == 2 ? row_len - 1 - j : j;
    dst[t] = mode == 1 ? 0.f : src[r + (int64_t)frame_len * jj];
  }
}

// ------------------------------------------------------------------ clip scales (A13 peak normalisation, folded FX scalars)
// scale = s / (|s| * max|x| + tiny(float32)): peak normalisation `a / max(|a| + tiny)` (event.py:535-536) of the clip
// s * x, where s is the product of the scalar FX in front of it (Gain, Invert); one workgroup per clip.
__device__ __forceinline__ float peak_scale_of(const float *__restrict__ x, int64_t n, float s, float *red) {
  float mx = 0.f, z0 = 0.f, z1 = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, fabsf(x[i]));
  block_reduce3(z0, mx, z1, red, threadIdx.x, 1024);
  return finite_f32((double)s / ((double)fabsf(s) * (double)mx + 1.17549435e-38));   // a silent clip under a gain of +12 dB or more: finite
}

// mode[e] 0: clip_scale[e] = prescale[e]; 1: the peak-normalising scale of clip e (events table gives offset / length)
__global__ __launch_bounds__(1024) void k_clip_scales(al_batch b, const float *__restrict__ prescale,
                                                      const int32_t *__restrict__ mode, float *__restrict__ out) {
  __shared__ float red[48];
  const int e = b.event0 + blockIdx.x;
  const al_event ev = b.events[e];
  const float s = prescale[e];
  if (mode[e] == 0) {
    if (threadIdx.x == 0) out[e] = s;
    return;
  }
  const float v = peak_scale_of(b.audio + ev.audio_off, ev.len, s, red);
  if (threadIdx.x == 0) out[e] = v;
}

__global__ __launch_bounds__(1024) void k_peak_scale(const float *__restrict__ x, int64_t n, float s, float *__restrict__ out) {
  __shared__ float red[48];
  const float v = peak_scale_of(x, n, s, red);
  if (threadIdx.x == 0) *out = v;
}

__global__ __launch_bounds__(256) void k_scale_matrix_rows(float *x, int64_t cols, const float *scale) {
  const float s = scale[blockIdx.y];
  float *row = x + (int64_t)blockIdx.y * cols;  // rows = channels of an ambience: far below grid.y's limit
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cols; i += (int64_t)gridDim.x * 256) row[i] *= s;
}

template <class T>
__global__ __launch_bounds__(256) void k_pack_irs(const T *src, float *dst, int len, int pitch) {
  const T *row = src + (int64_t)blockIdx.x * len;   // one workgroup per row: rows = C*N may exceed grid.y's 65535
  float *out = dst + (int64_t)blockIdx.x * pitch;
  for (int t = threadIdx.x; t < pitch; t += 256) out[t] = t < len ? (float)row[t] : 0.f;
}

// Ragged IRs (one 1-D array per (capsule, source), worldstate.py:2196-2253) -> zero-padded float32 rows of `pitch`.
template <class T>
__global__ __launch_bounds__(256) void k_pack_ragged(const T *__restrict__ src, const int64_t *__restrict__ offsets,
                                                     const int32_t *__restrict__ lens, float *__restrict__ dst, int pitch) {
  const int64_t row = blockIdx.x;
  const T *in = src + offsets[row];
  const int n = lens[row];
  float *out = dst + row * pitch;
  for (int t = threadIdx.x; t < pitch; t += 256) out[t] = t < n ? (float)in[t] : 0.f;
}

// Polyphase FIR resampling by up/down (scipy.signal.resample_poly semantics: zero-stuff by `up`, filter with h of
// 2*half+1 taps already scaled by `up`, keep every `down`-th sample): out[m] = sum_j x[j] * h[m*down - j*up + half].
__global__ __launch_bounds__(256) void k_resample_poly(const float *__restrict__ x, int64_t n_in, const float *__restrict__ h,
                                                       int half, int up, int down, float *__restrict__ out, int64_t n_out,
                                                       int64_t out_pitch) {
  const float *row = x + (int64_t)blockIdx.y * n_in;
  float *dst = out + (int64_t)blockIdx.y * out_pitch;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < out_pitch; m += (int64_t)gridDim.x * 256) {
    float acc = 0.f;
    if (m < n_out) {
      const int64_t c = m * down;  // position on the up-sampled grid
      // taps with 0 <= c - j*up + half <= 2*half  <=>  (c - half)/up <= j <= (c + half)/up
      int64_t j_lo = (c - half + up - 1) / up, j_hi = (c + half) / up;
      if (c - half < 0) j_lo = 0;
      if (j_hi > n_in - 1) j_hi = n_in - 1;
      for (int64_t j = j_lo; j <= j_hi; ++j) acc = fmaf(row[j], h[c - j * up + half], acc);
    }
    dst[m] = acc;
  }
}

// float -> PCM_16 as python-soundfile writes it: it enables SFC_SET_CLIPPING on every file it opens, so libsndfile converts
// with f2s_clip_array: scaled = x * 0x8000 (float); >= 0x7FFF -> 0x7FFF, <= -0x8000 -> -0x8000, else lrintf(scaled)
// (round half to even).  (Without clipping libsndfile scales by 0x7FFF instead; round 2 encoded that.)  soundfile is not in
// the build container, so this follows libsndfile's source, not a golden file: parity unpinned by definition.
__device__ __forceinline__ int16_t pcm16_of(float x) {
  const float scaled = x * 32768.0f;
  if (scaled >= 32767.0f) return (int16_t)32767;
  if (scaled <= -32768.0f) return (int16_t)-32768;
  return (int16_t)rintf(scaled);   // NaN never reaches here: non-finite scenes are refused before encoding
}

// (C, T) float32 scene -> (T, C) interleaved frames, the layout soundfile.write(audio.T) puts on disk (core.py:1840-1847).
// One workgroup per tile of 32 capsules x 64 samples through LDS: reads run along t, writes along c.
template <bool PCM16>
__global__ __launch_bounds__(256) void k_encode_frames(const float *__restrict__ scene, int n_capsules, int64_t n_samples,
                                                       void *__restrict__ out) {
  __shared__ float tile[32][65];
  const int64_t t0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 32; r += 4) {
    const int c = c0 + r + ty;
    tile[r + ty][tx] = (c < n_capsules && t0 + tx < n_samples) ? scene[(int64_t)c * n_samples + t0 + tx] : 0.f;
  }
  __syncthreads();
  // 16 bytes per store where the capsule count allows it (the destination may be page-locked HOST memory: every store
  // is then a PCIe write, and 2- or 4-byte stores make 64 / 128-byte packets)
  if (PCM16 && (n_capsules & 7) == 0) {
    const int tl = threadIdx.x >> 2, cg = (threadIdx.x & 3) * 8;   // one frame's 8 consecutive capsules
    const int64_t t = t0 + tl;
    if (t < n_samples && c0 + cg < n_capsules) {
      int16_t q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = pcm16_of(tile[cg + i][tl]);
      struct alignas(16) Frames8 { uint32_t w[4]; } v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v.w[i] = (uint32_t)(uint16_t)q[2 * i] | ((uint32_t)(uint16_t)q[2 * i + 1] << 16);
      *reinterpret_cast<Frames8 *>(reinterpret_cast<int16_t *>(out) + t * n_capsules + c0 + cg) = v;
    }
    return;
  }
  if (!PCM16 && (n_capsules & 3) == 0) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int j = threadIdx.x + 256 * pass, tl = j >> 3, cg = (j & 7) * 4;   // one frame's 4 consecutive capsules
      const int64_t t = t0 + tl;
      if (t < n_samples && c0 + cg < n_capsules)
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + t * n_capsules + c0 + cg) =
            make_float4(tile[cg][tl], tile[cg + 1][tl], tile[cg + 2][tl], tile[cg + 3][tl]);
    }
    return;
  }
  const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
#pragma unroll
  for (int tt = 0; tt < 64; tt += 8) {
    const int64_t t = t0 + tt + tl;
    const int c = c0 + cl;
    if (c < n_capsules && t < n_samples) {
      const float x = tile[cl][tt + tl];
      if (PCM16) {
        reinterpret_cast<int16_t *>(out)[t * n_capsules + c] = pcm16_of(x);
      } else {
        reinterpret_cast<float *>(out)[t * n_capsules + c] = x;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_wrap_copy(const float *src, int64_t m, float *dst, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) dst[t] = src[t % m];
}

}  // namespace al

// ====================================================================== C ABI
namespace {
thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_error(hipError_t e, const char *what) {
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return AL_E_HIP;
  }
  return AL_OK;
}

int check_launch(const char *what) { return check_error(hipGetLastError(), what); }

int check_batch(const al_batch *b) {
  if (!b) return fail(AL_E_BADARG, "null batch");
  if (b->struct_size != (int32_t)sizeof(al_batch) || b->abi_version != AL_ABI_VERSION)
    return fail(AL_E_BADARG, "al_batch was built against another version of audiblelight_hip.h (struct_size / abi_version)");
  if (b->log2_block < AL_MIN_LOG2_BLOCK || b->log2_block > AL_MAX_LOG2_BLOCK)
    return fail(AL_E_UNSUPPORTED, "log2_block must be in [10, 14]");
  if (b->n_capsules <= 0 || b->n_events < 0 || b->n_streams < 0) return fail(AL_E_BADARG, "bad batch sizes");
  if ((b->ir_stride_c & 3) || (b->ir_stride_n & 3)) return fail(AL_E_BADARG, "IR strides must be multiples of 4 floats");
  if (((uintptr_t)b->ir & 15) != 0) return fail(AL_E_BADARG, "IR base must be 16-byte aligned");
  if (b->n_emitters > 0 && b->n_partitions != ((b->ir_len + (1 << b->log2_block) - 1) >> b->log2_block))
    return fail(AL_E_BADARG, "n_partitions != ceil(ir_len / B)");
  if (b->hop <= 0) return fail(AL_E_BADARG, "hop must be positive");
  return AL_OK;
}

}  // namespace

// ONE description of what al_spectral_mac launches for a batch: al_spectral_mac launches from it, al_spectral_mac_variant
// reports it, so the parity tests' "which instantiation ran" assertion cannot drift from the launcher.
// Codes: tile kernel k_spectral_mac<KT,PT,VB,KSPLIT> = 1000000*KSPLIT + 10000*KT + 100*PT + VB; capsule-loop kernels
// 3120000 + 100*P + {1: k_spectral_mac_static<12,P,1>, 2: <12,P,2>, 3: k_spectral_mac_static_lds (P <= 12: <12,P,1>,
// 13..16: <12,ceil(P/2),2>)}; sliding-window kernel k_spectral_mac_moving<6,PT,1> = 600 + PT.
namespace {
enum MacStaticKind { MAC_STATIC_NONE = 0, MAC_STATIC_ONE = 1, MAC_STATIC_PAIR = 2, MAC_STATIC_LDS = 3, MAC_STATIC_LDS_UNITS = 4,
                     MAC_STATIC_GLDS = 5 };

struct MacPlan {
  int static_kind;    // MacStaticKind: which capsule-loop kernel takes the one-emitter events (NONE: the tile kernel does)
  int static_pt;      // its PT template argument
  dim3 static_grid;
  int static_threads;
  int static_code;    // what al_spectral_mac_variant reports for one-emitter events
  int tile_code;      // k_spectral_mac instantiation (0: not launched)
  int moving_code;    // k_spectral_mac_moving instantiation (0: not launched)
};

MacPlan plan_mac(const al_batch *b) {
  MacPlan m{};
  const bool wide_k = b->max_blocks > 8, wide_p = b->n_partitions > 4;
  const int bins = 1 << b->log2_block;
  // tile shapes: accumulators for up to 24 output blocks, 4 or 12 partition spectra in registers
  // (sweep of other shapes: profiles/r01_mac_variants.txt)
  if (wide_k && wide_p && bins >= 512) m.tile_code = 1121202;
  else if (wide_k) m.tile_code = 240401;
  else if (wide_p) m.tile_code = 81201;
  else m.tile_code = 80401;
  // moving events flagged by the planner (al_event.reserved == 1: every stream has n_j <= AL_SPARSE_MAX_NJ)
  if (b->n_streams > b->n_events && b->n_partitions <= AL_SPARSE_MAX_PARTITIONS)
    m.moving_code = 100 * AL_SPARSE_MAX_NJ + (b->n_partitions <= 12 ? 12 : AL_SPARSE_MAX_PARTITIONS);
  m.static_code = m.tile_code;
  if (al::static_mac_active(*b)) {
    const int P = b->n_partitions;
    // enough workgroups to fill the chip: split the capsule loop for small batches
    const int n_ktiles = (b->max_blocks + 11) / 12, base = (bins / 512) * n_ktiles * b->n_events;
    int n_cs = 1;
    while (n_cs < b->n_capsules && base * n_cs < 1024) n_cs *= 2;
    if (n_cs > b->n_capsules) n_cs = b->n_capsules;
    // flags bit 12 (A/B switch): one k-tile per workgroup; bit 13 (A/B switch): register version beyond 24 blocks
    const bool pair = n_ktiles > 1 && !(b->flags & (1 << 12));
    const int n_pairs = (n_ktiles + 1) / 2;
    if (P > 12) {
      // 13..16 partitions: two units of ceil(P/2) per capsule, 17..21: three units of ceil(P/3); always through LDS, always two
      // k-tiles per workgroup.  Fed by LDS-DMA (k_spectral_mac_static_glds: no staging registers, so the 32-block signal window
      // of 21 partitions fits; profiles/r03_p24_ab.txt) when the batch has an all-zero block for the rows past an odd count,
      // else (13..16 only) by the register-staged ring of k_spectral_mac_static_lds.
      const int units = P > 16 ? 3 : 2;
      m.static_kind = b->hspec_zero_block >= 0 ? MAC_STATIC_GLDS : MAC_STATIC_LDS_UNITS;
      m.static_pt = (P + units - 1) / units;
      m.static_grid = dim3(bins / 512, n_pairs, b->n_events * n_cs);
      m.static_threads = 512;
    } else if (pair && n_pairs > 1 && !(b->flags & (1 << 13))) {
      m.static_kind = MAC_STATIC_LDS;
      m.static_pt = P;
      m.static_grid = dim3(bins / 512, n_pairs, b->n_events * n_cs);
      m.static_threads = 512;
    } else if (pair) {
      m.static_kind = MAC_STATIC_PAIR;
      m.static_pt = P;
      m.static_grid = dim3(bins / 512, n_pairs, b->n_events * n_cs);
      m.static_threads = 512;
    } else {
      m.static_kind = MAC_STATIC_ONE;
      m.static_pt = P;
      m.static_grid = dim3(bins / 512, n_ktiles, b->n_events * n_cs);
      m.static_threads = 256;
    }
    // flags bit 14 (A/B switch): at most 12 partitions through the LDS-DMA kernel too (2 % slower there than the register /
    // register-staged versions: the accumulate of short IRs is not short of flight time, profiles/r03_p24_ab.txt)
    if ((b->flags & (1 << 14)) && (m.static_kind == MAC_STATIC_PAIR || m.static_kind == MAC_STATIC_LDS)) m.static_kind = MAC_STATIC_GLDS;
    m.static_code = 3120000 + 100 * P + (m.static_kind == MAC_STATIC_ONE ? 1 : m.static_kind == MAC_STATIC_PAIR ? 2 :
                                         m.static_kind == MAC_STATIC_GLDS ? 4 : 3);
    if (b->flags & AL_FLAG_ONLY_STATIC) m.tile_code = m.moving_code = 0;   // no event is left for the other kernels
  }
  return m;
}

template <int PT>
void launch_mac_static(const MacPlan &m, const al_batch *b, hipStream_t stream) {
  switch (m.static_kind) {
    case MAC_STATIC_ONE:
      hipLaunchKernelGGL((al::k_spectral_mac_static<12, PT, 1>), m.static_grid, dim3(256), 0, stream, *b);
      break;
    case MAC_STATIC_PAIR:
      hipLaunchKernelGGL((al::k_spectral_mac_static<12, PT, 2>), m.static_grid, dim3(512), 0, stream, *b);
      break;
    case MAC_STATIC_GLDS:
      hipLaunchKernelGGL((al::k_spectral_mac_static_glds<12, PT>), m.static_grid, dim3(512), 0, stream, *b);
      break;
    default:
      hipLaunchKernelGGL((al::k_spectral_mac_static_lds<12, PT>), m.static_grid, dim3(512), 0, stream, *b);
      break;
  }
}
}  // namespace

extern "C" {

const char *al_last_error(void) { return g_err; }
int al_abi_version(void) { return AL_ABI_VERSION; }

int64_t al_twiddle_bytes(int log2_block) {
  if (log2_block < AL_MIN_LOG2_BLOCK || log2_block > AL_MAX_LOG2_BLOCK) return -1;
  return (int64_t)sizeof(float) * 2 * ((int64_t)1 << log2_block);
}

int al_twiddle_init(float *twiddle, int log2_block, al_stream_t stream) {
  if (!twiddle || al_twiddle_bytes(log2_block) < 0) return fail(AL_E_BADARG, "bad twiddle arguments");
  const int m = 1 << log2_block;
  hipLaunchKernelGGL(al::k_twiddle_init, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<float2 *>(twiddle), m);
  return check_launch("k_twiddle_init");
}

int al_ir_spectra(const al_batch *b, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (b->n_emitters <= 0) return AL_OK;
  return check_error(al::launch_ir_spectra(b, (hipStream_t)stream), "k_ir_spectra");
}

int al_emitter_gains(const al_batch *b, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (b->n_emitters <= 0) return AL_OK;
  hipLaunchKernelGGL(al::k_emitter_gains, dim3(b->n_emitters), dim3(64), 0, (hipStream_t)stream, *b, 0, 0);
  return check_launch("k_emitter_gains");
}

int al_emitter_norm_sums(const al_batch *b, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (b->n_emitters <= 0) return AL_OK;
  hipLaunchKernelGGL(al::k_emitter_gains, dim3(b->n_emitters), dim3(64), 0, (hipStream_t)stream, *b, 1, 0);
  return check_launch("k_emitter_gains(sums)");
}

int al_emitter_gains_from_sums(const al_batch *b, int32_t total_capsules, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (total_capsules < b->n_capsules) return fail(AL_E_BADARG, "total_capsules < n_capsules");
  if (b->n_emitters <= 0) return AL_OK;
  hipLaunchKernelGGL(al::k_emitter_gains, dim3(b->n_emitters), dim3(64), 0, (hipStream_t)stream, *b, 2, total_capsules);
  return check_launch("k_emitter_gains(from sums)");
}

int al_forward_spectra(const al_batch *b, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  return check_error(al::launch_forward_spectra(b, (hipStream_t)stream), "k_forward_spectra");
}

int al_signal_spectra(const al_batch *b, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (b->n_streams <= 0 || b->max_nj <= 0) return AL_OK;
  return check_error(al::launch_signal_spectra(b, (hipStream_t)stream), "k_signal_spectra");
}

int al_spectral_mac_variant(const al_batch *b, int32_t *static_code, int32_t *moving_code) {
  if (int rc = check_batch(b)) return rc;
  if (!static_code || !moving_code) return fail(AL_E_BADARG, "null output");
  const MacPlan m = plan_mac(b);
  *static_code = m.static_code;
  *moving_code = m.moving_code;
  return AL_OK;
}

int al_spectral_mac(const al_batch *b, al_stream_t stream_) {
  if (int rc = check_batch(b)) return rc;
  if (b->n_events <= 0 || b->n_emitters <= 0) return AL_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const MacPlan m = plan_mac(b);
  const int bins = 1 << b->log2_block;
  if (m.static_kind == MAC_STATIC_GLDS && b->n_partitions > 12) {
    const int P = b->n_partitions;
    if (P <= 14) hipLaunchKernelGGL((al::k_spectral_mac_static_glds<12, 7, 2>), m.static_grid, dim3(512), 0, stream, *b);
    else if (P <= 16) hipLaunchKernelGGL((al::k_spectral_mac_static_glds<12, 8, 2>), m.static_grid, dim3(512), 0, stream, *b);
    else if (P <= 18) hipLaunchKernelGGL((al::k_spectral_mac_static_glds<12, 6, 3>), m.static_grid, dim3(512), 0, stream, *b);
    else hipLaunchKernelGGL((al::k_spectral_mac_static_glds<12, 7, 3, true, 2>), m.static_grid, dim3(512), 0, stream, *b);
    if (int rc = check_launch("k_spectral_mac_static_glds")) return rc;
  } else if (m.static_kind == MAC_STATIC_LDS_UNITS) {
    if (m.static_pt == 7) hipLaunchKernelGGL((al::k_spectral_mac_static_lds<12, 7, 2>), m.static_grid, dim3(512), 0, stream, *b);
    else hipLaunchKernelGGL((al::k_spectral_mac_static_lds<12, 8, 2>), m.static_grid, dim3(512), 0, stream, *b);
    if (int rc = check_launch("k_spectral_mac_static_lds")) return rc;
  } else if (m.static_kind != MAC_STATIC_NONE) {
    switch (m.static_pt) {   // the partition tile IS the partition count: no masked partitions in the loop
      case 1: launch_mac_static<1>(m, b, stream); break;
      case 2: launch_mac_static<2>(m, b, stream); break;
      case 3: launch_mac_static<3>(m, b, stream); break;
      case 4: launch_mac_static<4>(m, b, stream); break;
      case 5: launch_mac_static<5>(m, b, stream); break;
      case 6: launch_mac_static<6>(m, b, stream); break;
      case 7: launch_mac_static<7>(m, b, stream); break;
      case 8: launch_mac_static<8>(m, b, stream); break;
      case 9: launch_mac_static<9>(m, b, stream); break;
      case 10: launch_mac_static<10>(m, b, stream); break;
      case 11: launch_mac_static<11>(m, b, stream); break;
      case 12: launch_mac_static<12>(m, b, stream); break;
      default: return fail(AL_E_BADARG, "capsule-loop accumulate: bad partition tile");
    }
    if (int rc = check_launch("k_spectral_mac_static")) return rc;
  }
#define AL_MAC(KT_, PT_, VB_) \
  case 10000 * KT_ + 100 * PT_ + VB_: \
    hipLaunchKernelGGL((al::k_spectral_mac<KT_, PT_, VB_>), dim3(bins / (256 * VB_), b->n_capsules, b->n_events), dim3(256), 0, \
                       stream, *b); \
    break
#define AL_MAC_KS(KT_, PT_, VB_) \
  case 1000000 + 10000 * KT_ + 100 * PT_ + VB_: \
    hipLaunchKernelGGL((al::k_spectral_mac<KT_, PT_, VB_, true>), \
                       dim3(bins / (256 * VB_), b->n_capsules * ((b->max_blocks + KT_ - 1) / KT_), b->n_events), dim3(256), 0, \
                       stream, *b); \
    break
  switch (m.tile_code) {
    case 0: break;   // AL_FLAG_ONLY_STATIC: every event went through the capsule loop
    AL_MAC_KS(12, 12, 2);
    AL_MAC(8, 12, 1);
    AL_MAC(24, 4, 1);
    AL_MAC(8, 4, 1);
    default: return fail(AL_E_UNSUPPORTED, "unknown spectral MAC variant");
  }
#undef AL_MAC
#undef AL_MAC_KS
  if (int rc = check_launch("k_spectral_mac")) return rc;
  if (m.moving_code) {
    const dim3 grid(bins / 256, b->n_capsules, b->n_events);
    if (m.moving_code % 100 == 12)
      hipLaunchKernelGGL((al::k_spectral_mac_moving<AL_SPARSE_MAX_NJ, 12, 1>), grid, dim3(256), 0, stream, *b);
    else
      hipLaunchKernelGGL((al::k_spectral_mac_moving<AL_SPARSE_MAX_NJ, AL_SPARSE_MAX_PARTITIONS, 1>), grid, dim3(256), 0,
                         stream, *b);
    return check_launch("k_spectral_mac_moving");
  }
  return AL_OK;
}

int al_block_synthesis(const al_batch *b, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (b->n_events <= 0 || b->max_blocks <= 0) return AL_OK;
  return check_error(al::launch_block_synthesis(b, (hipStream_t)stream), "k_block_synthesis");
}

int al_event_levels(const al_batch *b, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (b->n_events <= 0) return AL_OK;
  hipLaunchKernelGGL(al::k_event_levels, dim3(b->n_events), dim3(64), 0, (hipStream_t)stream, *b, 0, 0);
  return check_launch("k_event_levels");
}

int al_event_stats(const al_batch *b, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (b->n_events <= 0) return AL_OK;
  hipLaunchKernelGGL(al::k_event_levels, dim3(b->n_events), dim3(64), 0, (hipStream_t)stream, *b, 1, 0);
  return check_launch("k_event_levels(stats)");
}

int al_event_levels_from_stats(const al_batch *b, int32_t total_capsules, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (total_capsules < b->n_capsules) return fail(AL_E_BADARG, "total_capsules < n_capsules");
  if (b->n_events <= 0) return AL_OK;
  hipLaunchKernelGGL(al::k_event_levels, dim3(b->n_events), dim3(64), 0, (hipStream_t)stream, *b, 2, total_capsules);
  return check_launch("k_event_levels(law)");
}

int al_render_batch(const al_batch *b, al_stream_t stream) {
  int rc;
  if ((rc = al_forward_spectra(b, stream))) return rc;
  if ((rc = al_emitter_gains(b, stream))) return rc;
  if ((rc = al_spectral_mac(b, stream))) return rc;
  if ((rc = al_block_synthesis(b, stream))) return rc;
  return al_event_levels(b, stream);
}

int al_mixdown(const al_mix *m, al_stream_t stream) {
  if (!m) return fail(AL_E_BADARG, "null mixdown descriptor");
  if (m->struct_size != (int32_t)sizeof(al_mix) || m->abi_version != AL_ABI_VERSION)
    return fail(AL_E_BADARG, "al_mix was built against another version of audiblelight_hip.h (struct_size / abi_version)");
  if (m->n_capsules <= 0 || m->n_samples <= 0 || m->tile <= 0) return fail(AL_E_BADARG, "bad mixdown arguments");
  if (m->tile != 4096) return fail(AL_E_BADARG, "mixdown tile must be 4096 samples");
  if (m->n_tiles != (m->n_samples + m->tile - 1) / m->tile) return fail(AL_E_BADARG, "n_tiles != ceil(n_samples / tile)");
  if (((uintptr_t)m->scene & 15) != 0) return fail(AL_E_BADARG, "scene buffer must be 16-byte aligned");
  if (m->ambience && (!m->ambience_scale || m->accumulate || ((uintptr_t)m->ambience & 15) != 0))
    return fail(AL_E_BADARG, "fused ambience needs ambience_scale, accumulate == 0 and a 16-byte aligned buffer");
  hipLaunchKernelGGL(al::k_mixdown, dim3(m->n_tiles, m->n_capsules), dim3(256), 0, (hipStream_t)stream, *m);
  return check_launch("k_mixdown");
}

int al_scale_rows(float *x, int64_t n, const float *scale, al_stream_t stream) {
  if (!x || !scale || n < 0) return fail(AL_E_BADARG, "bad scale arguments");
  if (n == 0) return AL_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(al::k_scale, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, n, scale);
  return check_launch("k_scale");
}

int al_scale_rows_f64(float *x, int64_t n, const double *scale, al_stream_t stream) {
  if (!x || !scale || n < 0) return fail(AL_E_BADARG, "bad scale arguments");
  if (n == 0) return AL_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(al::k_scale_d, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, n, scale);
  return check_launch("k_scale_d");
}

int al_clip_scales(const al_batch *b, const float *prescale, const int32_t *mode, al_stream_t stream) {
  if (int rc = check_batch(b)) return rc;
  if (!prescale || !mode || !b->clip_scale) return fail(AL_E_BADARG, "clip_scales needs prescale, mode and al_batch.clip_scale");
  if (b->n_events <= 0) return AL_OK;
  hipLaunchKernelGGL(al::k_clip_scales, dim3(b->n_events), dim3(1024), 0, (hipStream_t)stream, *b, prescale, mode,
                     const_cast<float *>(b->clip_scale));
  return check_launch("k_clip_scales");
}

int al_peak_scale(const float *x, int64_t n, float prescale, float *scale_out, al_stream_t stream) {
  if (!x || !scale_out || n <= 0) return fail(AL_E_BADARG, "bad peak_scale arguments");
  hipLaunchKernelGGL(al::k_peak_scale, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, prescale, scale_out);
  return check_launch("k_peak_scale");
}

int al_axpy(float *y, const float *x, const float *a_dev, int64_t n, al_stream_t stream) {
  if (!x || !y || !a_dev || n < 0) return fail(AL_E_BADARG, "bad axpy arguments");
  if (n == 0) return AL_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(al::k_axpy, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, y, x, a_dev, n);
  return check_launch("k_axpy");
}

int64_t al_row_stats_partials(int32_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return 4 * (int64_t)rows * ((cols + al::ROW_CHUNK - 1) / al::ROW_CHUNK);
}

int al_row_stats(const float *x, int32_t rows, int64_t cols, float *partials, double *out, al_stream_t stream) {
  if (!x || !partials || !out || rows <= 0 || cols <= 0) return fail(AL_E_BADARG, "bad row_stats arguments");
  const int nchunks = (int)((cols + al::ROW_CHUNK - 1) / al::ROW_CHUNK);
  if ((int64_t)nchunks * rows > 0x7fffffff) return fail(AL_E_BADARG, "row_stats: too many (row, chunk) pairs");
  hipLaunchKernelGGL(al::k_row_stats, dim3((unsigned)(nchunks * rows)), dim3(256), 0, (hipStream_t)stream, x, cols, partials);
  if (int rc = check_launch("k_row_stats")) return rc;
  hipLaunchKernelGGL(al::k_row_stats_final, dim3(rows), dim3(64), 0, (hipStream_t)stream, partials, nchunks, out);
  return check_launch("k_row_stats_final");
}

int al_fx_apply(int op, const float *src, float *dst, int64_t n, const float *params, const int32_t *iparams,
                al_stream_t stream) {
  if (!src || !dst || n <= 0) return fail(AL_E_BADARG, "bad fx arguments");
  if (op < AL_FX_GAIN || op > AL_FX_DEEMPH) return fail(AL_E_UNSUPPORTED, "unknown fx op");
  const bool out_of_place = (op == AL_FX_REVERSE || op == AL_FX_PREEMPH || op == AL_FX_DEEMPH);
  if (out_of_place && src == dst) return fail(AL_E_BADARG, "this fx op needs dst != src");
  if ((op == AL_FX_PREEMPH || op == AL_FX_DEEMPH) && n < 2) return fail(AL_E_BADARG, "emphasis filters need n >= 2");
  al::FxArgs a{op, params ? params[0] : 0.f, 0, 0, AL_FADE_NONE, AL_FADE_NONE};
  if (op == AL_FX_FADE) {
    if (!iparams) return fail(AL_E_BADARG, "fade needs iparams");
    a.n_in = iparams[0]; a.n_out = iparams[1]; a.shape_in = iparams[2]; a.shape_out = iparams[3];
  }
  if (op == AL_FX_DEEMPH) {
    hipLaunchKernelGGL(al::k_fx_deemph, dim3(1), dim3(1024), 0, (hipStream_t)stream, src, dst, n, a.p0);
    return check_launch("k_fx_deemph");
  }
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(al::k_fx_pointwise, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, n, a);
  return check_launch("k_fx_pointwise");
}

int al_fx_frame_shuffle(const float *src, float *dst, int64_t n, int32_t frame_len, int32_t row_len,
                        const int32_t *rows, int32_t n_rows, al_stream_t stream) {
  if (!src || !dst || !rows || n <= 0 || frame_len <= 0 || row_len <= 0 || n_rows <= 0 || src == dst)
    return fail(AL_E_BADARG, "bad frame_shuffle arguments");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(al::k_frame_shuffle, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, n, frame_len, row_len, rows, n_rows);
  return check_launch("k_frame_shuffle");
}

// ---- arbitrary-length inverse real FFT (ambience)
namespace {
struct BigPlan {
  int64_t len;       // complex transform length (n/2 for even n, n for odd n)
  bool bluestein;
  int64_t L;         // power-of-two Bluestein length (0 if unused)
};

int64_t strip_small_factors(int64_t m) {
  for (int f : {2, 3, 5, 7})
    while (m % f == 0) m /= f;
  return m;
}

BigPlan big_plan(int64_t n) {
  BigPlan p;
  p.len = (n & 1) ? n : n / 2;
  p.bluestein = strip_small_factors(p.len) != 1;
  p.L = 0;
  if (p.bluestein) {
    p.L = 1;
    while (p.L < 2 * p.len - 1) p.L <<= 1;
  }
  return p;
}

// Stockham passes over `rows` series of `len` points; returns the buffer holding the result.
float2 *big_fft(float2 *a, float2 *b, int rows, int64_t len, int dir, hipStream_t stream) {
  int64_t ns = 1, rest = len;
  float2 *in = a, *out = b;
  while (rest > 1) {
    int r = 0;
    for (int cand : {4, 2, 3, 5, 7})
      if (rest % cand == 0) { r = cand; break; }
    const dim3 grid((unsigned)((len / r + 255) / 256), rows);
    switch (r) {
      case 2: hipLaunchKernelGGL((al::k_big_pass<2>), grid, dim3(256), 0, stream, in, out, len, ns, dir); break;
      case 3: hipLaunchKernelGGL((al::k_big_pass<3>), grid, dim3(256), 0, stream, in, out, len, ns, dir); break;
      case 4: hipLaunchKernelGGL((al::k_big_pass<4>), grid, dim3(256), 0, stream, in, out, len, ns, dir); break;
      case 5: hipLaunchKernelGGL((al::k_big_pass<5>), grid, dim3(256), 0, stream, in, out, len, ns, dir); break;
      default: hipLaunchKernelGGL((al::k_big_pass<7>), grid, dim3(256), 0, stream, in, out, len, ns, dir); break;
    }
    ns *= r;
    rest /= r;
    float2 *t = in; in = out; out = t;
  }
  return in;
}
}  // namespace

int64_t al_noise_workspace_floats(int32_t rows, int64_t n) {
  if (rows <= 0 || n <= 0) return 0;
  const BigPlan p = big_plan(n);
  const int64_t per = p.bluestein ? p.L : p.len;
  return 2 * (2 * (int64_t)rows * per) + (p.bluestein ? 2 * 2 * p.L + 2 * (int64_t)rows * p.len : 0);
}

namespace {
// irfft of rows x (n/2+1) shaped draws; zr == nullptr: the draws come from the device generator under `seed`
int noise_irfft(const float *zr, const float *zi, uint64_t seed, const float *shape, int32_t rows, int64_t n, float inv_sigma,
                float *out, float *workspace, hipStream_t st) {
  const BigPlan p = big_plan(n);
  const int64_t per = p.bluestein ? p.L : p.len;
  float2 *a = reinterpret_cast<float2 *>(workspace);
  float2 *b = a + (int64_t)rows * per;
  const dim3 g_len((unsigned)((p.len + 255) / 256), rows);
  const float2 *z;
  if (!p.bluestein) {
    hipLaunchKernelGGL(al::k_noise_pack, g_len, dim3(256), 0, st, zr, zi, seed, shape, n, a);
    z = big_fft(a, b, rows, p.len, +1, st);
  } else {
    float2 *kern = b + (int64_t)rows * per;           // 2 * L: chirp kernel and its ping-pong partner
    float2 *x = kern + 2 * p.L;                        // rows * len: packed spectrum / transform result
    const dim3 g_L((unsigned)((p.L + 255) / 256), rows), g_L1((unsigned)((p.L + 255) / 256), 1);
    hipLaunchKernelGGL(al::k_noise_pack, g_len, dim3(256), 0, st, zr, zi, seed, shape, n, x);
    hipLaunchKernelGGL(al::k_blue_kernel, g_L1, dim3(256), 0, st, p.len, p.L, +1, kern);
    const float2 *kspec = big_fft(kern, kern + p.L, 1, p.L, -1, st);
    hipLaunchKernelGGL(al::k_blue_pre, g_L, dim3(256), 0, st, (const float2 *)x, p.len, p.L, +1, a);
    float2 *A = big_fft(a, b, rows, p.L, -1, st);
    hipLaunchKernelGGL(al::k_blue_mul, g_L, dim3(256), 0, st, A, kspec, p.L);
    float2 *other = (A == a) ? b : a;
    const float2 *y = big_fft(A, other, rows, p.L, +1, st);
    hipLaunchKernelGGL(al::k_blue_post, g_len, dim3(256), 0, st, y, p.len, p.L, +1, x);
    z = x;
  }
  hipLaunchKernelGGL(al::k_noise_unpack, g_len, dim3(256), 0, st, z, n, inv_sigma / (float)p.len, out);
  return check_launch("al_noise_irfft");  // hipGetLastError keeps the first failure of the sequence until it is read
}
}  // namespace

int al_noise_irfft(const float *zr, const float *zi, const float *shape, int32_t rows, int64_t n, float inv_sigma,
                   float *out, float *workspace, al_stream_t stream) {
  if (!zr || !zi || !shape || !out || !workspace || rows <= 0 || n <= 0) return fail(AL_E_BADARG, "bad noise arguments");
  return noise_irfft(zr, zi, 0, shape, rows, n, inv_sigma, out, workspace, (hipStream_t)stream);
}

int al_noise_irfft_seeded(uint64_t seed, const float *shape, int32_t rows, int64_t n, float inv_sigma, float *out,
                          float *workspace, al_stream_t stream) {
  if (!out || !workspace || rows <= 0 || n <= 0) return fail(AL_E_BADARG, "bad noise arguments");
  return noise_irfft(nullptr, nullptr, seed, shape, rows, n, inv_sigma, out, workspace, (hipStream_t)stream);
}

int al_normal_fill(float *out, int64_t n, uint64_t seed, uint32_t tag, float scale, al_stream_t stream) {
  if (!out || n <= 0) return fail(AL_E_BADARG, "bad normal_fill arguments");
  if (((uintptr_t)out & 15) != 0) return fail(AL_E_BADARG, "normal_fill: output must be 16-byte aligned");
  const int64_t blocks = (n / 4 + 256) / 256;
  hipLaunchKernelGGL(al::k_normal_fill, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, out, n,
                     seed, tag, scale);
  return check_launch("k_normal_fill");
}

int al_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
  if (!counter || !key || !out) return fail(AL_E_BADARG, "bad philox arguments");
  const al::Philox4 p = al::philox4x32_10(counter[0], counter[1], counter[2], counter[3], key[0], key[1]);
  out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = p.w;
  return AL_OK;
}

int al_ambience_scales(const double *row_stats, int32_t rows, int64_t cols, float ref_db, int32_t normalize, float *scales,
                       al_stream_t stream) {
  if (!row_stats || !scales || rows <= 0 || rows > 1024 || cols <= 0) return fail(AL_E_BADARG, "bad ambience_scales arguments");
  hipLaunchKernelGGL(al::k_ambience_scales, dim3(1), dim3(64), 0, (hipStream_t)stream, row_stats, rows, cols, ref_db, normalize, scales);
  return check_launch("k_ambience_scales");
}

int al_axpy_rows(float *y, const float *x, const float *a_dev, int32_t rows, int64_t cols, al_stream_t stream) {
  if (!x || !y || !a_dev || rows <= 0 || rows > 65535 || cols <= 0) return fail(AL_E_BADARG, "bad axpy_rows arguments");
  const int64_t blocks = (cols + 255) / 256;
  hipLaunchKernelGGL(al::k_axpy_rows, dim3((unsigned)(blocks < 2048 ? blocks : 2048), rows), dim3(256), 0, (hipStream_t)stream, y, x,
                     a_dev, cols);
  return check_launch("k_axpy_rows");
}

// ---- STFT-domain intermediates of the moving path (A7), reference signatures kept in audiblelight_amd/synthesize.py
namespace {
constexpr int64_t MAX_GRID_ROWS = 32768;  // series per launch group (grid.y limit is 65535)

// Complex FFT of `rows` series of `len` points for ANY len (the reference's numpy rfft / irfft take any fft_size, synthesize.py:135,263):
// Stockham passes where len factors into 2, 3, 5 and 7, Bluestein's chirp-z on a power-of-two length otherwise (the pieces of
// al_bigfft.h the ambience synthesis uses).  `x` holds the input, `y` is a second rows x len buffer, `tmp` any_fft_tmp_floats(rows, len)
// floats (none for smooth lengths).  Returns the buffer that holds the transform (x or y; always y on the Bluestein path).
int64_t bluestein_length(int64_t len) {
  int64_t L = 1;
  while (L < 2 * len - 1) L <<= 1;
  return L;
}
int64_t any_fft_tmp_floats(int64_t rows, int64_t len) {
  if (strip_small_factors(len) == 1) return 0;
  const int64_t L = bluestein_length(len);
  return 2 * (2 * rows * L + 2 * L);   // two rows x L ping-pong buffers, the chirp kernel and its partner
}
const float2 *any_fft(float2 *x, float2 *y, float *tmp, int rows, int64_t len, int dir, hipStream_t st) {
  if (strip_small_factors(len) == 1) return big_fft(x, y, rows, len, dir, st);
  const int64_t L = bluestein_length(len);
  float2 *a = reinterpret_cast<float2 *>(tmp), *b = a + (int64_t)rows * L, *kern = b + (int64_t)rows * L;
  const dim3 g_len((unsigned)((len + 255) / 256), rows), g_L((unsigned)((L + 255) / 256), rows), g_L1((unsigned)((L + 255) / 256), 1);
  hipLaunchKernelGGL(al::k_blue_kernel, g_L1, dim3(256), 0, st, len, L, dir, kern);
  const float2 *kspec = big_fft(kern, kern + L, 1, L, -1, st);
  hipLaunchKernelGGL(al::k_blue_pre, g_L, dim3(256), 0, st, (const float2 *)x, len, L, dir, a);
  float2 *A = big_fft(a, b, rows, L, -1, st);
  hipLaunchKernelGGL(al::k_blue_mul, g_L, dim3(256), 0, st, A, kspec, L);
  const float2 *conv = big_fft(A, (A == a) ? b : a, rows, L, +1, st);
  hipLaunchKernelGGL(al::k_blue_post, g_len, dim3(256), 0, st, conv, len, L, dir, y);
  return y;
}
}  // namespace

int64_t al_stft_workspace_floats(int64_t series, int32_t fft_size) {
  if (series <= 0 || fft_size <= 0) return 0;
  const int64_t group = series < MAX_GRID_ROWS ? series : MAX_GRID_ROWS;
  return 2 * 2 * group * (int64_t)fft_size + any_fft_tmp_floats(group, fft_size);  // ping-pong complex buffers for one launch group (+ Bluestein's)
}

int al_stft(const float *y, int64_t rows, int64_t n, int32_t fft_size, int32_t win_size, int32_t hop_size, float *spec,
            float *workspace, al_stream_t stream) {
  // fft_size < win_size is legal in the reference: rfft(frames, n=fft_size) crops every windowed frame to its first fft_size samples
  if (!y || !spec || !workspace || rows <= 0 || n <= 0 || win_size <= 0 || hop_size <= 0 || win_size < hop_size || fft_size <= 0)
    return fail(AL_E_BADARG, "bad stft arguments");
  hipStream_t st = (hipStream_t)stream;
  const int n_frames = 2 * (int)((n + 2 * (int64_t)hop_size - 1) / (2 * (int64_t)hop_size)) + 1;
  const int64_t series = rows * n_frames;
  const int64_t group = series < MAX_GRID_ROWS ? series : MAX_GRID_ROWS;
  float2 *a = reinterpret_cast<float2 *>(workspace);
  float2 *b = a + group * fft_size;
  float *tmp = reinterpret_cast<float *>(b + group * fft_size);
  for (int64_t s0 = 0; s0 < series; s0 += MAX_GRID_ROWS) {
    const int g = (int)(series - s0 < MAX_GRID_ROWS ? series - s0 : MAX_GRID_ROWS);
    const dim3 grid((unsigned)((fft_size + 255) / 256), g);
    hipLaunchKernelGGL(al::k_stft_pack, grid, dim3(256), 0, st, y, n, n_frames, fft_size, win_size, hop_size, s0, a);
    const float2 *z = any_fft(a, b, tmp, g, fft_size, -1, st);
    hipLaunchKernelGGL(al::k_stft_take_half, grid, dim3(256), 0, st, z, fft_size, s0, reinterpret_cast<float2 *>(spec));
  }
  return check_launch("al_stft");
}

int al_tv_stft_mac(const float *s_audio, const float *s_ir, const float *w_ir, int32_t n_frames, int32_t n_frames_ir,
                   int32_t n_freq, int32_t n_ch, int32_t n_irs, float *out, al_stream_t stream) {
  if (!s_audio || !s_ir || !w_ir || !out || n_frames <= 0 || n_frames_ir <= 0 || n_freq <= 0 || n_ch <= 0 || n_irs <= 0)
    return fail(AL_E_BADARG, "bad tv_stft_mac arguments");
  // one launch per MAX_GRID_ROWS output frames (grid.y limit): a clip of any length (hop 16 at 44.1 kHz passes 65 535 frames after 23 s)
  for (int32_t f0 = 0; f0 < n_frames; f0 += MAX_GRID_ROWS) {
    const int g = n_frames - f0 < MAX_GRID_ROWS ? n_frames - f0 : MAX_GRID_ROWS;
    const dim3 grid((unsigned)(((int64_t)n_freq * n_ch + 255) / 256), g);
    hipLaunchKernelGGL(al::k_tv_stft_mac, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2 *>(s_audio),
                       reinterpret_cast<const float2 *>(s_ir), w_ir, f0, n_frames_ir, n_freq, n_ch, n_irs,
                       reinterpret_cast<float2 *>(out));
  }
  return check_launch("k_tv_stft_mac");
}

int64_t al_istft_workspace_floats(int32_t n_frames, int32_t n_ch, int32_t fft_size) {
  if (n_frames <= 0 || n_ch <= 0 || fft_size <= 0) return 0;
  const int64_t series = (int64_t)n_frames * n_ch;
  return 2 * 2 * series * fft_size + any_fft_tmp_floats(series < MAX_GRID_ROWS ? series : MAX_GRID_ROWS, fft_size);
}

int al_istft_ola(const float *spatial_stft, int32_t n_frames, int32_t n_freq, int32_t n_ch, int32_t fft_size,
                 int32_t win_size, int32_t hop_size, float *out, float *workspace, al_stream_t stream) {
  if (!spatial_stft || !out || !workspace || n_frames <= 0 || n_ch <= 0 || fft_size <= 0 || win_size <= 0 || hop_size <= 0)
    return fail(AL_E_BADARG, "bad istft arguments");
  if (n_freq != fft_size / 2 + 1) return fail(AL_E_BADARG, "istft: n_freq must be fft_size / 2 + 1");
  if ((int64_t)n_frames * hop_size <= win_size) return fail(AL_E_BADARG, "istft: no output samples");
  hipStream_t st = (hipStream_t)stream;
  const int64_t series = (int64_t)n_frames * n_ch;
  float2 *a = reinterpret_cast<float2 *>(workspace), *b = a + series * fft_size;
  float *tmp = reinterpret_cast<float *>(b + series * fft_size);
  const float2 *frames = nullptr;
  for (int64_t s0 = 0; s0 < series; s0 += MAX_GRID_ROWS) {
    const int g = (int)(series - s0 < MAX_GRID_ROWS ? series - s0 : MAX_GRID_ROWS);
    const dim3 grid((unsigned)((fft_size + 255) / 256), g);
    hipLaunchKernelGGL(al::k_istft_pack, grid, dim3(256), 0, st, reinterpret_cast<const float2 *>(spatial_stft), n_freq, n_ch,
                       fft_size, s0, a + s0 * fft_size);
    const float2 *z = any_fft(a + s0 * fft_size, b + s0 * fft_size, tmp, g, fft_size, +1, st);
    frames = (z == a + s0 * fft_size) ? a : b;  // every group ends in the same buffer (same pass count; Bluestein: always the second)
  }
  const int64_t total = ((int64_t)n_frames * hop_size - win_size) * n_ch;
  hipLaunchKernelGGL(al::k_istft_ola, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, frames, n_frames, n_ch, fft_size,
                     win_size, hop_size, out);
  return check_launch("al_istft_ola");
}

int al_scale_matrix_rows(float *x, int32_t rows, int64_t cols, const float *scale, al_stream_t stream) {
  if (!x || !scale || rows <= 0 || rows > 65535 || cols <= 0) return fail(AL_E_BADARG, "bad scale_matrix_rows arguments");
  const int64_t blocks = (cols + 255) / 256;
  hipLaunchKernelGGL(al::k_scale_matrix_rows, dim3((unsigned)(blocks < 2048 ? blocks : 2048), rows), dim3(256), 0,
                     (hipStream_t)stream, x, cols, scale);
  return check_launch("k_scale_matrix_rows");
}

int al_pack_irs_f64(const double *src, float *dst, int64_t rows, int32_t len, int32_t dst_pitch, al_stream_t stream) {
  if (!src || !dst || rows <= 0 || len <= 0 || dst_pitch < len || (dst_pitch & 3) || rows > 0x7fffffff)
    return fail(AL_E_BADARG, "bad pack_irs arguments");
  hipLaunchKernelGGL(al::k_pack_irs<double>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, src, dst, len, dst_pitch);
  return check_launch("k_pack_irs<double>");
}

int al_pack_irs_f32(const float *src, float *dst, int64_t rows, int32_t len, int32_t dst_pitch, al_stream_t stream) {
  if (!src || !dst || src == dst || rows <= 0 || len <= 0 || dst_pitch < len || (dst_pitch & 3) || rows > 0x7fffffff)
    return fail(AL_E_BADARG, "bad pack_irs arguments");
  hipLaunchKernelGGL(al::k_pack_irs<float>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, src, dst, len, dst_pitch);
  return check_launch("k_pack_irs<float>");
}

int al_pack_ragged_irs(const void *src, int32_t src_is_f64, const int64_t *offsets, const int32_t *lens, int64_t rows,
                       int32_t dst_pitch, float *dst, al_stream_t stream) {
  if (!src || !offsets || !lens || !dst || rows <= 0 || rows > 0x7fffffff || dst_pitch <= 0 || (dst_pitch & 3))
    return fail(AL_E_BADARG, "bad pack_ragged_irs arguments");
  if (src_is_f64)
    hipLaunchKernelGGL(al::k_pack_ragged<double>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const double *>(src), offsets, lens, dst, dst_pitch);
  else
    hipLaunchKernelGGL(al::k_pack_ragged<float>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const float *>(src), offsets, lens, dst, dst_pitch);
  return check_launch("k_pack_ragged");
}

int al_resample_poly(const float *x, int32_t rows, int64_t n_in, const float *taps, int32_t half_len, int32_t up, int32_t down,
                     float *out, int64_t n_out, int64_t out_pitch, al_stream_t stream) {
  if (!x || !taps || !out || rows <= 0 || rows > 65535 || n_in <= 0 || half_len < 0 || up <= 0 || down <= 0 || n_out <= 0 ||
      out_pitch < n_out)
    return fail(AL_E_BADARG, "bad resample_poly arguments");
  const int64_t blocks = (out_pitch + 255) / 256;
  hipLaunchKernelGGL(al::k_resample_poly, dim3((unsigned)(blocks < 4096 ? blocks : 4096), rows), dim3(256), 0, (hipStream_t)stream,
                     x, n_in, taps, half_len, up, down, out, n_out, out_pitch);
  return check_launch("k_resample_poly");
}

int al_encode_frames(const float *scene, int32_t n_capsules, int64_t n_samples, int32_t format, void *out, al_stream_t stream) {
  if (!scene || !out || n_capsules <= 0 || n_samples <= 0 || (format != AL_FRAMES_F32 && format != AL_FRAMES_PCM16))
    return fail(AL_E_BADARG, "bad encode_frames arguments");
  const bool vector_stores = format == AL_FRAMES_PCM16 ? (n_capsules & 7) == 0 : (n_capsules & 3) == 0;
  if (vector_stores && ((uintptr_t)out & 15) != 0) return fail(AL_E_BADARG, "encode_frames: output must be 16-byte aligned");
  const dim3 grid((unsigned)((n_samples + 63) / 64), (unsigned)((n_capsules + 31) / 32));
  if (format == AL_FRAMES_PCM16)
    hipLaunchKernelGGL((al::k_encode_frames<true>), grid, dim3(256), 0, (hipStream_t)stream, scene, n_capsules, n_samples, out);
  else
    hipLaunchKernelGGL((al::k_encode_frames<false>), grid, dim3(256), 0, (hipStream_t)stream, scene, n_capsules, n_samples, out);
  return check_launch("k_encode_frames");
}

int al_wrap_copy(const float *src, int64_t m, float *dst, int64_t n, al_stream_t stream) {
  if (!src || !dst || m <= 0 || n <= 0 || src == dst) return fail(AL_E_BADARG, "bad wrap_copy arguments");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(al::k_wrap_copy, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                     (hipStream_t)stream, src, m, dst, n);
  return check_launch("k_wrap_copy");
}

}  // extern "C"
