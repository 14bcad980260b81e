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
// STATUS (round 4, profiles/r04b_cfg3_fused_first_version_ab.txt, r04c_moving_fused_phase_timing.txt): parity-green on the
// MI355X against the oracle in every regime, but SLOWER than the path over stored spectra: 6.9 ms + 1.3 ms for the energy-only
// forward pass against 3.5 + 2.7 ms on cfg3 (8.8 vs 6.9 ms per scene).  Timing builds show why: the phases of a round ADD UP --
// raw loads 2.7, transforms 1.6, products 1.5, stage 0.6, loop skeleton 1.0 ms -- because with 136 KB of LDS and 256 VGPRs
// only ONE 512-thread workgroup fits a CU and all its waves are in the same phase between barriers.  A second version
// (signal blocks parked in LDS, transform image as the stage, next round's samples requested a phase ahead behind LDS-only
// barriers; commit "k_moving_fused v3 experiment", profiles/r04d_*, r04e_*) measured 9.4 ms.  Opt-in: AL_FUSED_MOVING=1.
//
// The IR energies of normalize_irs (synthesize.py:404-428) come from the forward kernel in its energy-only mode
// (al_batch.emitter_parts[n] = 0): one read-only pass over the IR tensor; the gains must exist before the first product.
#pragma once
#include <hip/hip_runtime.h>

#include "al_common.h"
#include "al_fft.h"

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

// The Q-point transform input of tile `tile` for one partition: z[m] at n = tg + 128 m.  `ir` points at the partition's first
// sample, `remaining` = samples left in the IR from there (zeros beyond); `live` = false gives zeros without touching memory.
template <int TILE>
__device__ __forceinline__ void fold_partition(const float *__restrict__ ir, int remaining, bool live, int tg, float2 wb,
                                               float2 (&z)[16]) {
  constexpr int Q = 2048;
  constexpr FoldFactors ff{};
  constexpr float R2 = 0.70710678118654752440f;
  if (!live) {
#pragma unroll
    for (int m = 0; m < 16; ++m) z[m] = make_float2(0.f, 0.f);
    return;
  }
  const bool whole = remaining >= 4 * Q;   // group-uniform
  const int last = remaining - 1;
  if constexpr (TILE == 0) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int t = 2 * (tg + 128 * m);
      float2 lo, hi;
      if (whole) {
        lo = *reinterpret_cast<const float2 *>(ir + t);
        hi = *reinterpret_cast<const float2 *>(ir + t + 2 * Q);
      } else {
        const float a0 = ir[min(t, last)], a1 = ir[min(t + 1, last)], b0 = ir[min(t + 2 * Q, last)], b1 = ir[min(t + 2 * Q + 1, last)];
        lo = make_float2(t <= last ? a0 : 0.f, t + 1 <= last ? a1 : 0.f);
        hi = make_float2(t + 2 * Q <= last ? b0 : 0.f, t + 2 * Q + 1 <= last ? b1 : 0.f);
      }
      z[m] = make_float2(lo.x + hi.x, lo.y + hi.y);
    }
  } else {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int n = tg + 128 * m;
      float h0, h1, h2, h3;
      if (whole) {
        h0 = ir[n]; h1 = ir[n + Q]; h2 = ir[n + 2 * Q]; h3 = ir[n + 3 * Q];
      } else {
        const float a0 = ir[min(n, last)], a1 = ir[min(n + Q, last)], a2 = ir[min(n + 2 * Q, last)], a3 = ir[min(n + 3 * Q, last)];
        h0 = n <= last ? a0 : 0.f; h1 = n + Q <= last ? a1 : 0.f; h2 = n + 2 * Q <= last ? a2 : 0.f; h3 = n + 3 * Q <= last ? a3 : 0.f;
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
  __shared__ float2 img[NG][G::LDS_ELEMS];
  __shared__ float2 stage[NG][Q];
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
    for (int l0 = 0; l0 < ev.n_streams; l0 += 64) {
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
        Quad4 x[NJW];
#pragma unroll
        for (int jj = 0; jj < NJW; ++jj) {
          x[jj] = load4(X + (int64_t)(t.w + min(jj, nj - 1)) * B);
          const float sc = jj < nj ? gn : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) x[jj].s[i] = make_float2(x[jj].s[i].x * sc, x[jj].s[i].y * sc);
        }
        const float *irn = b.ir + (int64_t)c * b.ir_stride_c + (int64_t)t.z * b.ir_stride_n;
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
            float2 z[16];
            const float *src = irn + (int64_t)min(p, P - 1) * B;
            const int remaining = b.ir_len - min(p, P - 1) * B;
            switch (tile) {                               // workgroup-uniform
              case 0: fold_partition<0>(src, remaining, live, tg, wb, z); break;
              case 1: fold_partition<1>(src, remaining, live, tg, wb, z); break;
              case 2: fold_partition<2>(src, remaining, live, tg, wb, z); break;
              default: fold_partition<3>(src, remaining, live, tg, wb, z); break;
            }
            fft_regs_to_regs<G, -1>(z, img[g], tw, tg);   // z[m] = Z[tg + 128 m]; its last LDS reads end with a barrier
            if (tile == 0) {
              real_unpack_store_regs<G>(z, img[g], tw.w0, tg, stage[g]);
            } else {
#pragma unroll
              for (int m = 0; m < 16; ++m) stage[g][tg + 128 * m] = z[m];
            }
            __syncthreads();
            static_for<4>([&](auto gg_c) {
              constexpr int gg = decltype(gg_c)::value, pp = 4 * r + gg;
              if (p0 + pp < pe) {                         // workgroup-uniform
                const Quad4 h = load4(&stage[gg][4 * tid]);
#pragma unroll
                for (int jj = 0; jj < NJW; ++jj) {
                  if (jj >= nj) continue;                 // workgroup-uniform: blocks past the stream's last are not multiplied
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
            // no barrier here: the next write of the stage lies behind the barriers of the next round's transform passes
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

// Which instantiation takes a batch: NJW = 5 (cfg3: cross-fade windows of 2.9 blocks) in passes of 8 partitions, NJW = 6 in
// passes of 4 (the accumulator window is what fills the register file).
int moving_fused_code(const al_batch *b) { return 10000 + ((b->flags & AL_FLAG_FUSED_NJ5) ? 508 : 604); }

hipError_t launch_moving_fused(const al_batch *b, hipStream_t stream) {
  const int64_t pairs = (int64_t)b->n_capsules * b->n_events, ids = (pairs + 7) / 8 * 8 * 4;
  if (ids > 0x7fffffff) return hipErrorInvalidValue;
  const dim3 grid((unsigned)ids);
  if (b->flags & AL_FLAG_FUSED_NJ5) hipLaunchKernelGGL((k_moving_fused<5, 8>), grid, dim3(512), 0, stream, *b);
  else hipLaunchKernelGGL((k_moving_fused<6, 4>), grid, dim3(512), 0, stream, *b);
  return hipGetLastError();
}

}  // namespace al
