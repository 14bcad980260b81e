// Epilogue of the block-synthesis kernels (one-transform, split and quad-tile forms): the alias-free half of an inverse-transformed
// window goes to the event's (C, len) block of `spatial`, truncated / zero-padded to the clip (pad_or_truncate_audio,
// synthesize.py:590), with the partial sums of the level law (sum|x|, max|x|; synthesize.py:594-599).
#pragma once
#include <hip/hip_runtime.h>

#include "al_common.h"
#include "al_fft.h"

namespace al {

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

}  // namespace al
