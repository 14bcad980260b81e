// Moving events without the IR spectra round trip (included by al_transforms.hip; B = 8192 only).
//
// time_variant_convolution (audiblelight/synthesize.py:184-310) in the envelope form of DESIGN.md section 4 meets every IR
// partition spectrum H[n,c,p] exactly ONCE: IR n of a moving event is heard only through the 2-5 signal blocks of its
// cross-fade window.  The four-kernel path nevertheless writes all of them (cfg3: 12.6 GB) and reads them back (12.6 GB) --
// 25 of that scene's 36 GB.  Here the sliding-window accumulate of k_spectral_mac_moving makes its own H from the raw IR
// samples, so H never exists in memory.
//
// What makes that possible is the QUAD layout of the spectra (QuadSlots, al_fft.h): a block's B = 4Q slots are four tiles
// of Q = 2048 slots, each a plain Q-point complex FFT of a folded and twisted copy of the 2B-sample window
// (decimation in frequency, applied once to each half of the split layout of al_split.h).  For an IR partition
// h[0..B) (the window's second half is zero), with h0 = h[n], h1 = h[n + Q], h2 = h[n + 2Q], h3 = h[n + 3Q], n < Q,
// w = e^{-i pi n / B}, c8 = e^{-i pi / 4}:
//   T0 = W[4i]   (i < Q; slot 0 packs W[0], W[B]):  rFFT_{2Q}(a), a[t] = h[t] + h[t + 2Q], t < 2Q, as the Q-point complex
//                                                   FFT of a[2n] + i a[2n+1] followed by the real-FFT unpacking
//   T1 = W[8i+2]:  FFT_Q( ((h0 - h2) - i (h1 - h3)) w^2 )
//   T2 = W[8i+1]:  FFT_Q( ((h0 - i h2) + c8 (h1 - i h3)) w )
//   T3 = W[8i+5]:  FFT_Q( ((h0 - i h2) - c8 (h1 - i h3)) w^5 )
// (the signal and output spectra take the same layout through the slot maps of the split kernels: al_split.h, QUAD = true).
//
// One workgroup of 512 threads owns (event, capsule, tile): a thread owns 4 slots of the tile for the accumulate (the
// sliding window of k_spectral_mac_moving: W = NJW + PTW - 1 output blocks in registers, the stream's NJW signal blocks in
// registers, pre-multiplied by the emitter gain of normalize_irs), and the 512 threads are four groups of 128 that transform
// four partitions at a time (FftGeom<11,16>, the in-LDS transform of al_fft.h, all groups in step at its barriers) into an
// LDS stage from which every thread reads its 4 slots of each.  Each tile reads the whole 32 KB partition: the four
// workgroups of one (event, capsule) are adjacent on one XCD (dispatch order, see the id mapping), so three of the four reads
// are L2 hits.  More than PTW partitions are walked in passes of PTW, the later passes adding to the blocks the earlier
// ones stored (read-modify-write of Y by the thread that wrote it).
//
// The IR energies of normalize_irs (synthesize.py:404-428) come from the forward kernel in its energy-only mode
// (al_batch.emitter_parts[n] = 0): one read-only pass over the IR tensor; the gains must exist before the first product.
#pragma once
#include <hip/hip_runtime.h>

#include "al_common.h"
#include "al_fft.h"

// Timing builds: bit 2 no products, bit 4 no look-ahead (every round requests its own samples).  0 = the product.
// (The first version's phase timings -- loads, transforms, products, stage: profiles/r04c_moving_fused_phase_timing.txt.)
#ifndef AL_MF_SKIP
#define AL_MF_SKIP 0
#endif

namespace al {

// a += x * h (complex) as two v_pk_fma_f32 (see cfma_packed in al_kernels.hip: no swizzled operand copies in registers)
__device__ __forceinline__ void cfma_pk(float2 &a, const float2 &x, const float2 &h) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f av = {a.x, a.y};
  const v2f xv = {x.x, x.y}, hv = {h.x, h.y};
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(av) : "v"(xv), "v"(hv));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(av) : "v"(xv), "v"(hv));
  a = make_float2(av.x, av.y);
#else
  cfma(a, x, h);
#endif
}

struct Quad4 {   // a thread's four slots of a tile
  float2 s[4];
};

// e^{-i pi m / 64}, m < 16: tw[tg + 128 m] = tw[tg] * this (table of B = 8192 points, 128 threads per transform)
struct FoldFactors {
  float c[16], s[16];
  constexpr FoldFactors() : c{}, s{} {
    for (int m = 0; m < 16; ++m) {
      c[m] = (float)ct_cospi(m, 64);
      s[m] = (float)ct_sinpi(m, 64);
    }
  }
};

// The raw samples a tile's fold needs from one partition, requested as early as possible (a round before they are folded):
// 64 floats per thread.  Tile 0 folds pairs (h[2n], h[2n+1]) and (h[2n + 2Q], h[2n + 2Q + 1]), n = tg + 128 m; tiles 1..3 fold
// h[n + Q j], j < 4.  `ir` points at the partition's first sample, `remaining` = samples left in the IR from there (the loads
// are clamped into the row, the fold zeroes what lies beyond); `live` = false touches no memory.
template <bool TILE0>
__device__ __forceinline__ void issue_raw(const float *__restrict__ ir, int remaining, bool live, int tg, float (&raw)[64]) {
  constexpr int Q = 2048;
  if (!live) return;
  const bool whole = remaining >= 4 * Q;   // group-uniform
  const int last = remaining - 1;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    if constexpr (TILE0) {
      const int t = 2 * (tg + 128 * m);
      if (whole) {
        const float2 lo = *reinterpret_cast<const float2 *>(ir + t), hi = *reinterpret_cast<const float2 *>(ir + t + 2 * Q);
        raw[4 * m] = lo.x; raw[4 * m + 1] = lo.y; raw[4 * m + 2] = hi.x; raw[4 * m + 3] = hi.y;
      } else {
        raw[4 * m] = ir[min(t, last)]; raw[4 * m + 1] = ir[min(t + 1, last)];
        raw[4 * m + 2] = ir[min(t + 2 * Q, last)]; raw[4 * m + 3] = ir[min(t + 2 * Q + 1, last)];
      }
    } else {
      const int n = tg + 128 * m;
#pragma unroll
      for (int j = 0; j < 4; ++j) raw[4 * m + j] = whole ? ir[n + Q * j] : ir[min(n + Q * j, last)];
    }
  }
}

// The Q-point transform input of tile TILE from those samples: z[m] at n = tg + 128 m.
template <int TILE>
__device__ __forceinline__ void fold_raw(const float (&raw)[64], int remaining, bool live, int tg, float2 wb, float2 (&z)[16]) {
  constexpr int Q = 2048;
  constexpr FoldFactors ff{};
  constexpr float R2 = 0.70710678118654752440f;
  if (!live) {
#pragma unroll
    for (int m = 0; m < 16; ++m) z[m] = make_float2(0.f, 0.f);
    return;
  }
  const bool whole = remaining >= 4 * Q;
  const int last = remaining - 1;
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    float h0 = raw[4 * m], h1 = raw[4 * m + 1], h2 = raw[4 * m + 2], h3 = raw[4 * m + 3];
    if constexpr (TILE == 0) {
      if (!whole) {
        const int t = 2 * (tg + 128 * m);
        h0 = t <= last ? h0 : 0.f; h1 = t + 1 <= last ? h1 : 0.f; h2 = t + 2 * Q <= last ? h2 : 0.f; h3 = t + 2 * Q + 1 <= last ? h3 : 0.f;
      }
      z[m] = make_float2(h0 + h2, h1 + h3);
    } else {
      const int n = tg + 128 * m;
      if (!whole) {
        h0 = n <= last ? h0 : 0.f; h1 = n + Q <= last ? h1 : 0.f; h2 = n + 2 * Q <= last ? h2 : 0.f; h3 = n + 3 * Q <= last ? h3 : 0.f;
      }
      const float2 w = cmul(wb, make_float2(ff.c[m], -ff.s[m]));    // e^{-i pi n / B}
      const float2 w2 = cmul(w, w);
      if constexpr (TILE == 1) {
        z[m] = cmul(make_float2(h0 - h2, h3 - h1), w2);
      } else {
        const float2 p = make_float2(h0, -h2);                            // h0 - i h2
        const float2 q = make_float2(R2 * (h1 - h3), -R2 * (h1 + h3));    // c8 (h1 - i h3)
        if constexpr (TILE == 2) z[m] = cmul(cadd(p, q), w);
        else z[m] = cmul(csub(p, q), cmul(cmul(w2, w2), w));
      }
    }
  }
}

template <int NJW, int PTW>
__global__ __launch_bounds__(512) void k_moving_fused(al_batch b) {
  using G = FftGeom<11, 16>;
  constexpr int Q = 2048, B = 8192, W = NJW + PTW - 1, NG = 4, ROUNDS = PTW / 4;
  static_assert(PTW % 4 == 0 && G::T == 128, "four 128-thread groups transform four partitions per round");
  __shared__ float2 img[NG][G::LDS_ELEMS];      // transform images; after the last pass: the partition's tile in natural order (padded)
  __shared__ float4 xlds[NJW * 2][512];          // the stream's signal blocks, each thread's own 4 slots (read back by their writer only)
  __shared__ int4 tab[64];    // {j_lo, n_j, emitter, xspec_base - xspec_block0}
  __shared__ float gains[64];
  const int tid = threadIdx.x, g = tid >> 7, tg0 = tid & 127;
  // workgroup id -> (pair, tile): ids go round-robin over the 8 XCDs, so id % 8 is the XCD; the four tiles of one
  // (event, capsule) pair sit on ONE XCD in consecutive dispatch slots (they read the same 32 KB partitions: one HBM read,
  // three L2 hits), consecutive pairs (capsules of one event: the same signal spectra) on neighbouring XCDs
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, tile = slot & 3, pair = (slot >> 2) * 8 + xcd;
  const int C = b.n_capsules;
  if (pair >= C * b.n_events) return;
  const int c = pair % C, e = pair / C;
  const al_event ev = b.events[b.event0 + e];
  if (ev.n_streams <= 1 || ev.reserved != 1) return;   // workgroup-uniform: static / dense events go through the other kernels
  const int K = ev.n_blocks, P = b.n_partitions;
  const float2 *table = reinterpret_cast<const float2 *>(b.twiddle);
  FftTwiddles<G> tw;
  load_fft_twiddles<G, -1, 4>(tw, table, tg0);
  const float2 wb0 = table[tg0];
  const float2 *__restrict__ X = reinterpret_cast<const float2 *>(b.xspec) + tile * Q + 4 * tid;
  float2 *__restrict__ Y = reinterpret_cast<float2 *>(b.yspec) + ((int64_t)(ev.yspec_base - b.yspec_block0) + (int64_t)c * K) * B + tile * Q + 4 * tid;
  const bool bin0 = tile == 0 && tid == 0;
  const bool bin0_wave = tile == 0 && tid < 64;   // wave-uniform: only that wave pays for the (DC, Nyquist) slot's select

  auto load4 = [](const float2 *p) {
    Quad4 v;
    const float4 a = *reinterpret_cast<const float4 *>(p), bq = *reinterpret_cast<const float4 *>(p + 2);
    v.s[0] = make_float2(a.x, a.y); v.s[1] = make_float2(a.z, a.w); v.s[2] = make_float2(bq.x, bq.y); v.s[3] = make_float2(bq.z, bq.w);
    return v;
  };
  auto store4 = [](float2 *p, const Quad4 &v) {
    stream_store<1>(reinterpret_cast<float4 *>(p), make_float4(v.s[0].x, v.s[0].y, v.s[1].x, v.s[1].y));
    stream_store<1>(reinterpret_cast<float4 *>(p + 2), make_float4(v.s[2].x, v.s[2].y, v.s[3].x, v.s[3].y));
  };
  auto zero4 = []() {
    Quad4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.s[i] = make_float2(0.f, 0.f);
    return v;
  };

  for (int p0 = 0; p0 < P; p0 += PTW) {
    const bool first = p0 == 0;     // later passes add to what the earlier ones stored
    Quad4 acc[W];
    int kbase = -1;                 // output block held in acc[0]; -1: the window has not been placed yet
    float raw[64];                  // the samples of the round after the one being transformed (issue_raw)
    bool prefetched = false;
    for (int l0 = 0; l0 < ev.n_streams; l0 += 64) {
      prefetched = false;           // the look-ahead stops at the end of a table chunk
      __syncthreads();
      if (tid < 64 && l0 + tid < ev.n_streams) {
        const al_stream st = b.streams[ev.stream0 + l0 + tid];
        tab[tid] = make_int4(st.j_lo, st.n_j, st.emitter, st.xspec_base - b.xspec_block0);
        gains[tid] = b.emitter_gain[st.emitter];
      }
      __syncthreads();
      const int nl = min(64, ev.n_streams - l0);
      for (int l = 0; l < nl; ++l) {
        const int4 t = tab[l];
        const int jlo = t.x, nj = t.y;
        if (nj <= 0) continue;
        // partitions of this IR that reach a block the event keeps (pad_or_truncate, synthesize.py:590): p < K - j_lo
        const int pe = min(min(P, K - jlo), p0 + PTW);
        if (pe <= p0) continue;                           // nothing of this stream in this pass (workgroup-uniform)
        const int anchor = jlo + p0;                      // first block this stream's partitions [p0, pe) can reach
        if (kbase < 0) {                                  // place the window
          kbase = first ? 0 : anchor;
#pragma unroll
          for (int w = 0; w < W; ++w) acc[w] = (first || kbase + w >= K) ? zero4() : load4(Y + (int64_t)(kbase + w) * B);
        }
        while (kbase < anchor) {                          // retire the blocks before it: they are complete for this pass
          if (kbase < K) store4(Y + (int64_t)kbase * B, acc[0]);
#pragma unroll
          for (int w = 0; w + 1 < W; ++w) acc[w] = acc[w + 1];
          acc[W - 1] = (first || kbase + W >= K) ? zero4() : load4(Y + (int64_t)(kbase + W) * B);
          ++kbase;
        }
        const float gn = gains[l];
        // the stream's signal blocks, times the emitter gain, parked in LDS: live in registers only during the products
        // (a thread reads back only what it wrote itself: no barrier)
#pragma unroll
        for (int jj = 0; jj < NJW; ++jj) {
          if (jj < nj) {                                  // workgroup-uniform
            const float2 *xp = X + (int64_t)(t.w + jj) * B;
            const float4 a = *reinterpret_cast<const float4 *>(xp), bq = *reinterpret_cast<const float4 *>(xp + 2);
            xlds[2 * jj][tid] = make_float4(a.x * gn, a.y * gn, a.z * gn, a.w * gn);
            xlds[2 * jj + 1][tid] = make_float4(bq.x * gn, bq.y * gn, bq.z * gn, bq.w * gn);
          }
        }
        const float *irc = b.ir + (int64_t)c * b.ir_stride_c;
        static_for<ROUNDS>([&](auto r_c) {
          constexpr int r = decltype(r_c)::value;
          if (p0 + 4 * r < pe) {                          // workgroup-uniform: the round has a live partition
            // everything DERIVED from the lane index and the twiddle set (addresses, fifteen powers per butterfly, the fold's
            // sixteen twists) is loop-invariant: left visible, hipcc hoists all of it out of the stream loop and spills
            const int tg = opaque_lane(tg0);
            float2 wb = wb0;
            make_opaque(wb);
            tw.hide_from_hoisting();
            const int p = p0 + 4 * r + g;                 // this group's partition
            const bool live = p < pe;                     // group-uniform
            const int remaining = b.ir_len - min(p, P - 1) * B;
            if (!prefetched) {                            // first round of a table chunk: nobody requested its samples yet
              const float *src = irc + (int64_t)t.z * b.ir_stride_n + (int64_t)min(p, P - 1) * B;
              if (tile == 0) issue_raw<true>(src, remaining, live, tg, raw);
              else issue_raw<false>(src, remaining, live, tg, raw);
            }
            float2 z[16];
            switch (tile) {                               // workgroup-uniform
              case 0: fold_raw<0>(raw, remaining, live, tg, wb, z); break;
              case 1: fold_raw<1>(raw, remaining, live, tg, wb, z); break;
              case 2: fold_raw<2>(raw, remaining, live, tg, wb, z); break;
              default: fold_raw<3>(raw, remaining, live, tg, wb, z); break;
            }
            // all passes through LDS: the last one leaves the transform in the image in natural order (slot i at pad(i)) AND in
            // z[m] = Z[tg + 128 m], and ends with a barrier -- the image IS the stage the products read
            FftPasses<G, -1, 0, false, true>::run(z, img[g], tw, tg);
            if (tile == 0) {
              // real-FFT unpacking of tile 0, in place: thread owning k = tg + 128 m (m < 8) reads Z[M - k] from the image and
              // writes slots k and M - k; nobody else reads or writes either, so no barrier in between (formulas: al_fft.h)
              constexpr int M = G::M, T = G::T, H = G::H;
              constexpr PackFactors<G> pf{};
              float2 *s_ = img[g];
#pragma unroll
              for (int m = 0; m < H; ++m) {
                const int k = tg + T * m;
                if (m == 0 && k == 0) {
                  const float2 z0 = z[0], zh = z[H];
                  s_[0] = make_float2(z0.x + z0.y, z0.x - z0.y);
                  s_[G::pad(M / 2)] = cconj(zh);
                } else {
                  const float2 wm = m == 0 ? tw.w0 : cmul(tw.w0, make_float2(pf.c[m], -pf.s[m]));
                  const float2 zk = z[m], zm = s_[G::pad(M - k)];
                  const float2 e_ = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                  const float2 d_ = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
                  const float2 wo = cmul(wm, make_float2(d_.y, -d_.x));
                  s_[G::pad(k)] = cadd(e_, wo);
                  s_[G::pad(M - k)] = cconj(csub(e_, wo));
                }
              }
            }
            // The NEXT round's samples are requested now and folded a whole accumulate phase later: with one workgroup per CU
            // and every wave in the same phase nothing else would hide their latency (profiles/r04c_moving_fused_phase_timing.txt:
            // the loads alone were 2.7 of the first version's 6.9 ms).  Next round = this stream's (r + 1) or the first round of
            // the next stream of this table chunk that has a partition in this pass.
            {
              int nemit = t.z, np_ = p0 + 4 * (r + 1) + g, npe = pe;
              bool found = (r + 1 < ROUNDS) && (p0 + 4 * (r + 1) < pe);
              if (!found) {
                for (int l2 = l + 1; l2 < nl; ++l2) {
                  const int4 t2 = tab[l2];
                  const int pe2 = min(min(P, K - t2.x), p0 + PTW);
                  if (t2.y > 0 && pe2 > p0) { found = true; nemit = t2.z; np_ = p0 + g; npe = pe2; break; }
                }
              }
              prefetched = found && !(AL_MF_SKIP & 16);
              if (prefetched) {
                const float *src = irc + (int64_t)nemit * b.ir_stride_n + (int64_t)min(np_, P - 1) * B;
                const int rem2 = b.ir_len - min(np_, P - 1) * B;
                const int tg2 = opaque_lane(tg0);
                if (tile == 0) issue_raw<true>(src, rem2, np_ < npe, tg2, raw);
                else issue_raw<false>(src, rem2, np_ < npe, tg2, raw);
              }
            }
            if (tile == 0) block_barrier<true>();         // tile 0's unpacking wrote the image after the last pass's barrier
            Quad4 x[NJW];
#pragma unroll
            for (int jj = 0; jj < NJW; ++jj) {
              if (jj < nj) {
                const float4 a = xlds[2 * jj][tid], bq = xlds[2 * jj + 1][tid];
                x[jj].s[0] = make_float2(a.x, a.y); x[jj].s[1] = make_float2(a.z, a.w);
                x[jj].s[2] = make_float2(bq.x, bq.y); x[jj].s[3] = make_float2(bq.z, bq.w);
              }
            }
            static_for<4>([&](auto gg_c) {
              constexpr int gg = decltype(gg_c)::value, pp = 4 * r + gg;
              if (p0 + pp < pe) {                         // workgroup-uniform
                Quad4 h;                                  // slots 4 tid .. 4 tid + 3 of partition gg: contiguous inside one padded 16-group
                const float2 *hp = &img[gg][4 * tid + (tid >> 2)];
#pragma unroll
                for (int i = 0; i < 4; ++i) h.s[i] = hp[i];
#pragma unroll
                for (int jj = 0; jj < NJW; ++jj) {
                  if ((AL_MF_SKIP & 4) || jj >= nj) continue;   // workgroup-uniform: blocks past the stream's last are not multiplied
                  if (bin0_wave) {                        // (DC, Nyquist) packed in slot 0 of tile 0: two real products in lane 0
                    const float2 a0 = acc[jj + pp].s[0];
                    cfma_pk(acc[jj + pp].s[0], x[jj].s[0], h.s[0]);
                    if (bin0) acc[jj + pp].s[0] = make_float2(fmaf(x[jj].s[0].x, h.s[0].x, a0.x), fmaf(x[jj].s[0].y, h.s[0].y, a0.y));
                  } else {
                    cfma_pk(acc[jj + pp].s[0], x[jj].s[0], h.s[0]);
                  }
                  cfma_pk(acc[jj + pp].s[1], x[jj].s[1], h.s[1]);
                  cfma_pk(acc[jj + pp].s[2], x[jj].s[2], h.s[2]);
                  cfma_pk(acc[jj + pp].s[3], x[jj].s[3], h.s[3]);
                }
              }
            });
            block_barrier<true>();                        // the images are free for the next round's first pass (LDS-only
                                                          // barriers throughout: the look-ahead loads stay in flight)
          }
        });
      }
    }
    if (kbase < 0) {                                      // no stream reached this pass
      if (first)
        for (int k = 0; k < K; ++k) store4(Y + (int64_t)k * B, zero4());
      continue;
    }
#pragma unroll
    for (int w = 0; w < W; ++w)
      if (kbase + w < K) store4(Y + (int64_t)(kbase + w) * B, acc[w]);
    if (first)                                            // blocks beyond the last window: no stream reaches them
      for (int k = kbase + W; k < K; ++k) store4(Y + (int64_t)k * B, zero4());
  }
}

// One instantiation: streams of at most 5 signal blocks (cfg3: cross-fade windows of 2.9 blocks), passes of 8 partitions.  The
// LDS budget (four transform images + the stream's signal blocks: 150 KB of 160) is what bounds NJW; longer streams stay
// on k_spectral_mac_moving over stored spectra.
#ifndef AL_MF_PTW5
#define AL_MF_PTW5 8   /* partitions per pass (A/B switch of the timing builds) */
#endif
int moving_fused_code(const al_batch *b) { (void)b; return 10000 + 500 + AL_MF_PTW5; }

hipError_t launch_moving_fused(const al_batch *b, hipStream_t stream) {
  const int64_t pairs = (int64_t)b->n_capsules * b->n_events, ids = (pairs + 7) / 8 * 8 * 4;
  if (ids > 0x7fffffff) return hipErrorInvalidValue;
  const dim3 grid((unsigned)ids);
  hipLaunchKernelGGL((k_moving_fused<5, AL_MF_PTW5>), grid, dim3(512), 0, stream, *b);
  return hipGetLastError();
}

}  // namespace al
