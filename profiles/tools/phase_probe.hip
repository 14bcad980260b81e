// profiles/r03_stream_probe2.txt: this box reads at 7.1 TB/s, writes at 6.4, but moves a MIX of both at 5.3-5.6.  Can software
// un-mix the traffic?  Every wave reads the chip-wide 100 MHz clock (s_memrealtime) and issues its loads only in even windows of
// 2^SHIFT ticks and its stores only in odd ones, so the whole chip alternates between a read phase and a write phase without any
// communication; the data waits in registers in between.  Copy kernel, 16 B per lane, U loads in flight per thread.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wait_phase(int want, int shift) {
  while ((int)((__builtin_amdgcn_s_memrealtime() >> shift) & 1) != want) __builtin_amdgcn_s_sleep(2);
}

// SHIFT < 0: no phasing (plain copy with the same structure)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4f *__restrict__ in, v4f *__restrict__ out, size_t n, int shift) {
  const size_t per = n / gridDim.x, base = (size_t)blockIdx.x * per;
  for (size_t i = base + threadIdx.x; i + (U - 1) * 256 < base + per; i += U * 256) {
    v4f v[U];
    if (shift >= 0) wait_phase(0, shift);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * 256) : in[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(v[u]));      // the data has arrived before the write window is awaited
    if (shift >= 0) wait_phase(1, shift);
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * 256); else out[i + u * 256] = v[u]; }
  }
}

template <int U, bool NT>
void run(const void *a, void *b, size_t bytes, int grid, int shift) {
  const size_t n = bytes / 16;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(256), 0, 0, (const v4f *)a, (v4f *)b, n, shift);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  if (shift < 0) printf("U=%2d nt=%d grid=%5d  unphased          copy %.2f TB/s\n", U, NT, grid, 2.0 * bytes / 1e9 / best);
  else printf("U=%2d nt=%d grid=%5d  window %6.2f us  copy %.2f TB/s\n", U, NT, grid, (1 << shift) * 0.01, 2.0 * bytes / 1e9 / best);
  fflush(stdout);
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  void *a, *b;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
  for (int grid : {2048, 4096}) {
    for (int shift : {-1, 6, 7, 8, 9, 10, 11}) {
      run<8, true>(a, b, bytes, grid, shift);
      run<16, true>(a, b, bytes, grid, shift);
      run<8, false>(a, b, bytes, grid, shift);
    }
  }
  return 0;
}
