// Counter-based normal draws on the device for the ambience noise (A12, ambience.py:271-375,160-165).
//
// The reference draws with numpy's default_rng(seed).normal (PCG64 + ziggurat), a sequential, data-dependent stream that
// cannot be reproduced in parallel; its own acceptance tests for the noise are statistical (tests/test_ambience.py:30-67)
// plus fixed-seed reproducibility (:70-76).  Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
// SC'11; known-answer vectors of Random123 in tests/test_host_logic.py) gives every element its own counter, so the draws
// are a pure function of (seed, tag, element index): reproducible for a seed, independent of the launch geometry.
// Box-Muller turns two uniforms into two normals; the radius uniform keeps all 32 bits near zero (tail to 6.7 sigma).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace al {

struct Philox4 {
  uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                          uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// two independent N(0, 1) from two 32-bit words
__device__ __forceinline__ float2 box_muller(uint32_t a, uint32_t b) {
  const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;   // (0, 1]: small a is exact, so the tail is resolved
  const float u2 = (float)b * 2.3283064365386963e-10f;            // [0, 1]
  const float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincospif(2.0f * u2, &s, &c);
  return make_float2(r * c, r * s);
}

// the four normals of counter block `i` under (seed, tag)
__device__ __forceinline__ float4 normal4(uint64_t seed, uint32_t tag, uint64_t i) {
  const Philox4 p = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), tag, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float2 a = box_muller(p.x, p.y), b = box_muller(p.z, p.w);
  return make_float4(a.x, a.y, b.x, b.y);
}

// out[i] = scale * N(0,1), element i = normal (i % 4) of counter block i / 4: four outputs per thread, one 16-byte store
__global__ __launch_bounds__(256) void k_normal_fill(float *__restrict__ out, int64_t n, uint64_t seed, uint32_t tag, float scale) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; 4 * q < n; q += (int64_t)gridDim.x * 256) {
    float4 v = normal4(seed, tag, (uint64_t)q);
    v = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    if (4 * q + 3 < n) {
      *reinterpret_cast<float4 *>(out + 4 * q) = v;
    } else {
      const float e[4] = {v.x, v.y, v.z, v.w};
      for (int j = 0; 4 * q + j < n; ++j) out[4 * q + j] = e[j];
    }
  }
}

// (zr, zi) of spectrum bin f of row `row`: the first Box-Muller pair of counter block row * bins + f under tag 1
__device__ __forceinline__ float2 spectrum_draw(uint64_t seed, int64_t row, int64_t bins, int64_t f) {
  const uint64_t i = (uint64_t)row * (uint64_t)bins + (uint64_t)f;
  const Philox4 p = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return box_muller(p.x, p.y);
}

}  // namespace al
