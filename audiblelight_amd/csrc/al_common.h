// Pieces shared by the two translation units of the HIP extension (al_transforms.hip: FFT kernels, built
// with -fno-slp-vectorize; al_kernels.hip: everything else + the C ABI, built with default vectorisation).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "../../include/audiblelight_hip.h"

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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
