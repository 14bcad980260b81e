// Fused accumulate + block synthesis for static events at B = 8192 (included by al_transforms.hip).
//
// Why: the output spectra Y (written by k_spectral_mac, read back by k_block_synthesis) are 45 % of a scene's HBM
// traffic.  Here one workgroup of 512 threads owns KT consecutive output blocks of one (event,
// capsule): it accumulates Y[k0 .. k0+KT) in registers (KT * 64 KiB = the register file of a CU is what bounds KT),
// hands each block through LDS into the transform layout, and runs the inverse real FFT + truncation + level
// statistics of k_block_synthesis on it.  Y never exists in memory.  What it costs: the partition spectra H of an (event,
// capsule) are read by K / KT workgroups instead of 2, and the signal spectra X by a sliding window per k-tile -- from
// L2 / Infinity Cache, since the workgroups of one (event, capsule) are adjacent in dispatch order.
//
// Phase A, "accumulate layout": thread t owns bins {2t + 1024 m, 2t + 1 + 1024 m : m < 8} (16-byte loads, a wave reads
// 1 KiB contiguous).  For one m the partitions are walked in order with a sliding window of KT signal blocks
// (X[k0+kk-p] for step p+1 is X[k0+kk-1-p] of step p), so a step costs ONE H load and ONE X load for KT complex FMAs per
// bin.  The whole (m, p) walk of a partition tile is one statically unrolled stream of loads, issued AL_FUSED_DEPTH
// loads ahead of their use through a register ring: with one workgroup per CU nothing else hides the L2 latency.
// Out-of-range blocks (k - p outside the clip, p beyond the last partition) are read from a zero block: the choice is
// workgroup-uniform, i.e. scalar address arithmetic, no per-lane masks.
// Phase B: per block, registers -> LDS image (natural order) -> the (k, M - k) pairs of the real-FFT packing in the
// transform layout -> inverse passes -> epilogue of k_block_synthesis.
#pragma once
#include <hip/hip_runtime.h>

#include "al_common.h"
#include "al_fft.h"

#ifndef AL_FUSED_DEPTH
#define AL_FUSED_DEPTH 12
#endif
#ifndef AL_FUSED_KT
#define AL_FUSED_KT 4   /* output blocks per workgroup: 32 * KT accumulator registers per thread */
#endif
#ifndef AL_FUSED_PK
#define AL_FUSED_PK 1
#endif

namespace al {

// acc (two bins: xy, zw) += x * h, complex
__device__ __forceinline__ void cfma_pair(float4 &acc, const float4 &x, const float4 &h) {
#if defined(__HIP_DEVICE_COMPILE__) && AL_FUSED_PK
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 a0 = {acc.x, acc.y}, a1 = {acc.z, acc.w};
  const v2 h0 = {h.x, h.y}, h0s = {h.y, h.x}, h1 = {h.z, h.w}, h1s = {h.w, h.z};
  const v2 x0r = {x.x, x.x}, x0i = {-x.y, x.y}, x1r = {x.z, x.z}, x1i = {-x.w, x.w};
  a0 = __builtin_elementwise_fma(x0r, h0, a0);
  a0 = __builtin_elementwise_fma(x0i, h0s, a0);
  a1 = __builtin_elementwise_fma(x1r, h1, a1);
  a1 = __builtin_elementwise_fma(x1i, h1s, a1);
  acc = make_float4(a0.x, a0.y, a1.x, a1.y);
#else
  acc.x = fmaf(x.x, h.x, acc.x);
  acc.x = fmaf(-x.y, h.y, acc.x);
  acc.y = fmaf(x.x, h.y, acc.y);
  acc.y = fmaf(x.y, h.x, acc.y);
  acc.z = fmaf(x.z, h.z, acc.z);
  acc.z = fmaf(-x.w, h.w, acc.z);
  acc.w = fmaf(x.z, h.w, acc.w);
  acc.w = fmaf(x.w, h.z, acc.w);
#endif
}

// Epilogue shared with k_block_synthesis: v[H..E-1] of every thread are the alias-free samples [B, 2B) of the window.
#ifndef AL_SHIFTED_PAIRS
#define AL_SHIFTED_PAIRS 1
#endif
template <class G>
__device__ __forceinline__ void synth_store_block(const float2 (&v)[G::E], float *__restrict__ out, const al_event &ev,
                                                  int tbase, bool pair_ok, int tid, float &asum, float &amax) {
  constexpr int M = G::M, T = G::T, H = G::H;
  if (pair_ok && tbase + M <= ev.valid_len) {  // interior block (workgroup-uniform): unconditional pair stores
    float *o = out + tbase + 2 * tid;
#pragma unroll
    for (int m = 0; m < H; ++m) {
      const float2 z = v[H + m];
      stream_store<16>(reinterpret_cast<float2 *>(o + 2 * T * m), z);
      asum += fabsf(z.x) + fabsf(z.y);
      amax = fmaxf(amax, fmaxf(fabsf(z.x), fabsf(z.y)));
    }
  } else if (AL_SHIFTED_PAIRS && !pair_ok && tbase + M <= ev.valid_len) {
    // interior block of a row that starts at an ODD float offset (clip lengths are rarely even: every other capsule row of
    // such an event): the 8-byte aligned pairs are (x[2i+1], x[2i+2]), the second taken from the next lane; a wave's run of
    // 128 samples is closed by one 4-byte store at each end.  (Two 4-byte stores per lane here cost 15 % of the kernel.)
    float *o = out + tbase + 2 * tid;
    const int lane = tid & 63;
#pragma unroll
    for (int m = 0; m < H; ++m) {
      const float2 z = v[H + m];
      const float nx = __shfl_down(z.x, 1, 64);
      if (lane == 0) o[2 * T * m] = z.x;
      if (lane != 63) stream_store<16>(reinterpret_cast<float2 *>(o + 2 * T * m + 1), make_float2(z.y, nx));
      else o[2 * T * m + 1] = z.y;
      asum += fabsf(z.x) + fabsf(z.y);
      amax = fmaxf(amax, fmaxf(fabsf(z.x), fabsf(z.y)));
    }
  } else {
#pragma unroll
    for (int m = 0; m < H; ++m) {
      const int i = tid + T * m;  // complex index inside the kept half
      const float2 z = v[H + m];
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
}

// FULL: the partition count is a multiple of PT (no ragged partition tile, no range check on H)
template <int KT, int PT, bool FULL>
__global__ __launch_bounds__(512, 2) void k_mac_synthesis(al_batch b) {
  using G = FftGeom<13, 16>;
  constexpr int M = G::M, T = G::T, H = G::H, E = G::E;
  static_assert(T == 512 && M == 8192, "the fused kernel is built for B = 8192");
  constexpr int NM = M / (2 * T);                 // 16-byte columns per thread (8)
  constexpr int NL = (KT - 1) + 2 * PT;           // loads per column: window prefill + (H, X) per partition
  constexpr int TOTAL = NM * NL;
  constexpr int D = AL_FUSED_DEPTH;
  __shared__ float2 s[G::LDS_ELEMS];
  __shared__ float red[48];
  const int tid = threadIdx.x;
  // Workgroup -> (event, capsule, k-tile), XCD-aware: workgroups are dealt round-robin over the 8 XCDs, each with its
  // own L2, so the k-tiles of one (event, capsule) -- which all read the same partition spectra -- take ids that are
  // congruent mod 8 and consecutive within that residue class: they run together on ONE XCD and H is fetched into one
  // L2 once.  (Placement is a speed matter only; nothing depends on it for correctness.)
  const int n_ktiles = (b.max_blocks + KT - 1) / KT;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int pair = (seq / n_ktiles) * 8 + xcd;
  if (pair >= b.n_capsules * b.n_events) return;
  const int c = pair % b.n_capsules, e = b.event0 + pair / b.n_capsules;
  const al_event ev = b.events[e];
  if (ev.n_streams != 1) return;                  // moving / tiled events: the unfused kernels
  const int K = ev.n_blocks, P = b.n_partitions;
  const int k0 = (seq % n_ktiles) * KT;
  if (k0 >= K) return;
  const al_stream st = b.streams[ev.stream0];

  // block indices inside the chunk-local spectra workspaces
  const int hblock0 = ((st.emitter - b.emitter0) * b.n_capsules + c) * P;
  const int xblock0 = st.xspec_base - b.xspec_block0 - st.j_lo;   // block of signal block j is xblock0 + j
  const int jlo = st.j_lo, jhi = st.j_lo + st.n_j;
  // X: one descriptor over the whole signal-spectra workspace including its trailing all-zero block (< 4 GiB, checked
  // by al_fused_supported); a block k - p outside the clip is read from the zero block -- a SCALAR choice of offset.
  // H: one descriptor over this (event, capsule)'s P partitions; only a ragged last partition tile (P % PT != 0) has
  // loads beyond it, which get a poisoned lane offset (hardware range check -> 0).
  const SpectraView xv(reinterpret_cast<const float2 *>(b.xspec), b.xspec_zero_block + 1, M);
  const SpectraView hv(reinterpret_cast<const float2 *>(b.hspec) + (int64_t)hblock0 * M, P, M);
  const int lane_bytes = tid * 16;
  constexpr int BLOCK_BYTES = M * 8, COL_BYTES = T * 16;

  float4 acc[KT][NM];
#pragma unroll
  for (int kk = 0; kk < KT; ++kk)
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[kk][m] = make_float4(0.f, 0.f, 0.f, 0.f);

#ifndef AL_FUSED_DEBUG
#define AL_FUSED_DEBUG 0   /* timing experiments only: 1 = skip phase A, 2 = skip phase B */
#endif
  for (int p0 = 0; p0 < (AL_FUSED_DEBUG == 1 ? 0 : P); p0 += PT) {
    // load `idx` of this partition tile (uniform block choice, per-lane column)
    auto load_at = [&](auto idx_c) -> float4 {
      constexpr int idx = decltype(idx_c)::value;
      constexpr int m = idx / NL, r = idx % NL;
      auto x_block = [&](int j) {                 // X[j], or the zero block when j is outside the clip's blocks
        const int blk = (j >= jlo && j < jhi) ? xblock0 + j : b.xspec_zero_block;
        return xv.load(blk * BLOCK_BYTES + m * COL_BYTES, lane_bytes);
      };
      if constexpr (r < KT - 1) {                 // window prefill: X[k0 - p0 + r + 1]
        return x_block(k0 - p0 + (r + 1));
      } else {
        constexpr int q = r - (KT - 1), pp = q / 2;
        if constexpr ((q & 1) == 0) {             // H[p0 + pp]
          const int poison = (FULL || p0 + pp < P) ? 0 : SPECTRA_POISON;
          return hv.load((p0 + pp) * BLOCK_BYTES + m * COL_BYTES, lane_bytes + poison);
        } else {                                  // the window's new block X[k0 - p0 - pp]
          return x_block(k0 - p0 - pp);
        }
      }
    };
    float4 ring[D];
    float4 xw[KT], hcur = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < KT; ++i) xw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    static_for<TOTAL + D>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      if constexpr (i >= D) {                     // consume load i - D
        constexpr int idx = i - D;
        constexpr int m = idx / NL, r = idx % NL;
        const float4 val = ring[idx % D];
        if constexpr (r < KT - 1) {
          xw[r + 1] = val;                        // slot of kk = r + 1 at step 0
        } else {
          constexpr int q = r - (KT - 1), pp = q / 2;
          if constexpr ((q & 1) == 0) {
            hcur = val;
          } else {
            xw[((KT - (pp % KT)) % KT)] = val;    // kk = 0 at step pp
            static_for<KT>([&](auto kk_c) {
              constexpr int kk = decltype(kk_c)::value;
              constexpr int slot = ((kk - pp) % KT + KT) % KT;
              cfma_pair(acc[kk][m], xw[slot], hcur);
            });
          }
        }
      }
      if constexpr (i < TOTAL) ring[i % D] = load_at(i_c);
      pipeline_fence();                           // keep the issue order: the scheduler would sink every load to its use
    });
  }

  FftTwiddles<G> tw;   // requested here: their latency runs under the bin-0 fix-up, and they do not occupy registers above
  load_fft_twiddles<G, 1>(tw, reinterpret_cast<const float2 *>(b.twiddle), tid);

  // bin 0 packs (DC, Nyquist): two real products instead of a complex one.  Lane l of wave 0 redoes partitions
  // l, l + 64, ... for the KT blocks, the sums are reduced over the wave and replace thread 0's first bin.
  if (tid < 64) {
    const float2 *X2 = reinterpret_cast<const float2 *>(b.xspec), *H2 = reinterpret_cast<const float2 *>(b.hspec);
    float2 part[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) part[kk] = make_float2(0.f, 0.f);
    for (int p = tid; p < P; p += 64) {
      const float2 h = H2[(int64_t)(hblock0 + p) * M];
#pragma unroll
      for (int kk = 0; kk < KT; ++kk) {
        const int j = k0 + kk - p;
        const bool ok = j >= jlo && j < jhi;
        const float2 x = X2[(int64_t)(xblock0 + (ok ? j : jlo)) * M];
        part[kk].x = fmaf(ok ? x.x : 0.f, h.x, part[kk].x);
        part[kk].y = fmaf(ok ? x.y : 0.f, h.y, part[kk].y);
      }
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        part[kk].x += __shfl_down(part[kk].x, off, 64);
        part[kk].y += __shfl_down(part[kk].y, off, 64);
      }
      if (tid == 0) {
        acc[kk][0].x = part[kk].x;
        acc[kk][0].y = part[kk].y;
      }
    }
  }

  // ---- phase B: one block at a time through LDS into the transform layout, inverse real FFT, epilogue
  const float scale = b.emitter_gain[st.emitter] / (float)M;   // normalize_irs scalar applied here (static events)
  float *out = b.spatial + ev.out_off + (int64_t)c * ev.len;
  const bool pair_ok = (((ev.out_off + (int64_t)c * ev.len) & 1) == 0);
#if AL_FUSED_DEBUG == 2
  {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kk = 0; kk < KT; ++kk)
      for (int m = 0; m < NM; ++m) { t.x += acc[kk][m].x; t.y += acc[kk][m].y; t.z += acc[kk][m].z; t.w += acc[kk][m].w; }
    reinterpret_cast<float4 *>(out + (int64_t)k0 * M)[tid] = t;
    return;
  }
#endif
  static_for<KT>([&](auto kk_c) {
    constexpr int kk = decltype(kk_c)::value;
    const int k = k0 + kk;
    if (k < K) {                                  // workgroup-uniform
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int f = 2 * tid + 2 * T * m;        // even, so f and f + 1 share a padding group
        s[G::pad(f)] = make_float2(acc[kk][m].x, acc[kk][m].y);
        s[G::pad(f) + 1] = make_float2(acc[kk][m].z, acc[kk][m].w);
      }
      __syncthreads();
      float2 yk[H], ym[H];
#pragma unroll
      for (int m = 0; m < H; ++m) {
        const int kb = tid + T * m;
        yk[m] = s[G::pad(kb)];
        ym[m] = s[G::pad(kb == 0 ? M / 2 : M - kb)];
      }
      __syncthreads();                            // the image is rewritten by the packing step
      float2 v[E];
      real_pack_finish<G>(yk, ym, v, s, tw.w0, tid, scale);
      fft_regs_to_regs<G, 1>(v, s, tw, tid);
      float asum = 0.f, amax = 0.f;
      synth_store_block<G>(v, out, ev, k * M, pair_ok, tid, asum, amax);
      float bad = isfinite(asum) ? 0.f : 1.f;     // a NaN or Inf anywhere makes the sum of magnitudes non-finite
      if (bad != 0.f) { asum = 0.f; amax = 0.f; }
      block_reduce3(asum, amax, bad, red, tid, T);
      if (tid == 0) {
        float *pp = b.partials + 4 * ((int64_t)ev.part_base + (int64_t)c * ev.n_blocks + k);
        pp[0] = asum;
        pp[1] = amax;
        pp[2] = bad;
        pp[3] = 0.f;
      }
      __syncthreads();                            // `red` and the LDS image are reused by the next block
    }
  });
}

}  // namespace al
