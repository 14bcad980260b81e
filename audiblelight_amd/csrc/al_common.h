// Pieces shared by the two translation units of the HIP extension (al_transforms.hip: FFT kernels, built
// with -fno-slp-vectorize; al_kernels.hip: everything else + the C ABI, built with default vectorisation).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "../../include/audiblelight_hip.h"

// ------------------------------------------------------------------ schedule perturbation (TEST BUILDS ONLY: -DAL_SHAKE=<seed>)
// The inline-asm paths of this library -- LDS-DMA with hand-counted s_waitcnt vmcnt(N), LDS-only barriers, look-ahead loads that stay
// in flight across barriers -- are invisible to the host-emulation build the sanitizers and the differential fuzz run on.  What
// protects them is that the result must not depend on the ORDER in which the waves of a workgroup reach their barriers.  A build with
// AL_SHAKE = 1, 2, ... (tests/shake.py; never the product library) makes every wave sleep a wave-, workgroup- and site-dependent
// number of cycles after every workgroup barrier of either kind, after every LDS-DMA issue and before every counted wait, so waves
// leave each barrier out of step by up to a few hundred cycles; tests/test_gpu_shake.py asserts that every kernel family renders bit for
// bit what the product library renders.  A missing barrier (round 4's al_quad16.h race) shows up as a mismatch there.
#ifndef AL_SHAKE
#define AL_SHAKE 0
#endif
namespace al {
__device__ __forceinline__ void shake(int site) {
#if AL_SHAKE && defined(__HIP_DEVICE_COMPILE__)
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned h = (wave + 1u) * 2654435761u ^ (blockIdx.x * 40503u + blockIdx.y * 9973u + blockIdx.z * 7919u + (unsigned)site * 97u + AL_SHAKE * 1013904223u);
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
#if AL_SHAKE == 3   /* wave 0 -- the wave that finishes block-wide reductions and publishes results -- is the LAST to move on, by about
                       10 us: longer than a load from HBM takes, so that a wave waiting for its look-ahead loads still overtakes it */
  const int n = (wave == 0 && (site == 102 || site == 104)) ? 48 : (int)(h & 1);   // after either kind of barrier
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
#elif AL_SHAKE == 4 /* ... and always the first */
  const int n = wave == 0 ? 0 : 3 + (int)(h & 3);
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(3);
#else
  const int n = (int)(h & 7);
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(3);   // 3 x 64 cycles each: up to ~1350 cycles of skew per site
#endif
#else
  (void)site;
#endif
}
}  // namespace al
#if AL_SHAKE && defined(__HIP_DEVICE_COMPILE__)
// every __syncthreads() of the kernel sources: the same barrier, with the waves shaken out of step on both sides of it
__device__ __forceinline__ void al_shaken_syncthreads() {
  al::shake(101);
  __syncthreads();
  al::shake(102);
}
#define __syncthreads() al_shaken_syncthreads()
#endif

namespace al {
// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Returns x unchanged but unknown to the optimiser.  Used on the lane index at the top of a loop over blocks: the
// lane-dependent addresses of the loop body would otherwise all be hoisted as loop invariants (two registers each)
// and overflow the register file.
__device__ __forceinline__ int opaque_lane(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
  return x;
}

// ------------------------------------------------------------------ streamed (touch-once) global accesses
// AL_NT is a bit mask of the sites that use non-temporal loads / stores (experiment switch, see profiles/r02_nt.txt):
//   1 Y store (accumulate)   2 Y load (synthesis)   4 H / X store (spectra kernels)   8 IR load
//   16 event audio store (synthesis)   64 scene store (mixdown)
// Default 81 = Y, event-audio and scene stores: written once, read by a LATER kernel after 1.5-3 GB of other traffic,
// so keeping them out of the caches is free; measured -3 % on the accumulate and the mixdown.  Non-temporal H / X stores
// and Y loads cost time (the spectra kernel's own partner workgroups and the synthesis re-use those lines).
#ifndef AL_NT
#define AL_NT 81
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef float al_v2f __attribute__((ext_vector_type(2)));
typedef float al_v4f __attribute__((ext_vector_type(4)));
#endif
template <int SITE>
__device__ __forceinline__ void stream_store(float2 *p, const float2 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((AL_NT & SITE) != 0) {
    al_v2f t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<al_v2f *>(p));
    return;
  }
#endif
  *p = v;
}
template <int SITE>
__device__ __forceinline__ void stream_store(float4 *p, const float4 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((AL_NT & SITE) != 0) {
    al_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<al_v4f *>(p));
    return;
  }
#endif
  *p = v;
}
template <int SITE>
__device__ __forceinline__ void stream_store(float *p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((AL_NT & SITE) != 0) {
    __builtin_nontemporal_store(v, p);
    return;
  }
#endif
  *p = v;
}
template <int SITE>
__device__ __forceinline__ float2 stream_load(const float2 *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((AL_NT & SITE) != 0) {
    const al_v2f t = __builtin_nontemporal_load(reinterpret_cast<const al_v2f *>(p));
    return make_float2(t.x, t.y);
  }
#endif
  return *p;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence: hipcc puts s_waitcnt vmcnt(0) in
// front of its s_barrier, which drains every global load in flight -- a kernel that requests data a phase ahead of its use
// (the quad-tile transforms of al_quad16.h: the next partition during this one's transforms) would wait for it at the very next barrier.  Here only
// the LDS counter is waited on; registers loaded from global memory are still guarded by the waits hipcc inserts at their use.
template <bool LDS_ONLY>
__device__ __forceinline__ void block_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (LDS_ONLY) {
    shake(103);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    shake(104);
    return;
  }
#endif
  __syncthreads();
}

__device__ __forceinline__ void pipeline_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// ------------------------------------------------------------------ block-wide reductions
// sum / max / sum over the workgroup; result valid in thread 0.  `scratch` holds 3 floats per wave.
__device__ __forceinline__ void block_reduce3(float &a_sum, float &b_max, float &c_sum, float *scratch, int tid,
                                              int nthreads) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a_sum += __shfl_down(a_sum, off, 64);
    b_max = fmaxf(b_max, __shfl_down(b_max, off, 64));
    c_sum += __shfl_down(c_sum, off, 64);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) {
    scratch[wave * 3 + 0] = a_sum;
    scratch[wave * 3 + 1] = b_max;
    scratch[wave * 3 + 2] = c_sum;
  }
  __syncthreads();
  if (tid == 0) {
    const int nw = (nthreads + 63) >> 6;
    for (int w = 1; w < nw; ++w) {
      a_sum += scratch[w * 3 + 0];
      b_max = fmaxf(b_max, scratch[w * 3 + 1]);
      c_sum += scratch[w * 3 + 2];
    }
  }
}

// launchers of the transform kernels (defined in al_transforms.hip); return hipGetLastError() of the launch
hipError_t launch_ir_spectra(const al_batch *b, hipStream_t stream);
hipError_t launch_signal_spectra(const al_batch *b, hipStream_t stream);
hipError_t launch_forward_spectra(const al_batch *b, hipStream_t stream);
hipError_t launch_block_synthesis(const al_batch *b, hipStream_t stream);

}  // namespace al
