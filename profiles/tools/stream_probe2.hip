// What is the best streaming rate this MI355X gives, and which access shape gets it?  (profiles/r01_stream_probe.txt topped out at
// 5.0 TB/s for a float4 copy and 5.8 for float2, MI355X_MICROARCH.md quotes 6.29 TB/s.)  Sweeps: bytes per lane (8 / 16), loads in
// flight per thread (1..8), workgroup placement (grid-stride over everything vs one contiguous chunk per workgroup), grid size, and
// non-temporal hints; copy, read-only and write-only.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0 copy, 1 read only, 2 write only.  CHUNKED: workgroup w owns elements [w * per, (w + 1) * per); else grid-stride.
template <class T, int U, int MODE, bool NT, bool CHUNKED>
__global__ __launch_bounds__(256) void k(const T *__restrict__ in, T *__restrict__ out, size_t n, float *sink) {
  const size_t per = n / gridDim.x;
  const size_t base = CHUNKED ? (size_t)blockIdx.x * per : 0, limit = CHUNKED ? base + per : n;
  const size_t stride = CHUNKED ? 256 : (size_t)gridDim.x * 256;
  T acc = {};
  for (size_t i = base + (CHUNKED ? 0 : (size_t)blockIdx.x * 256) + threadIdx.x; i + (U - 1) * stride < limit; i += U * stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE != 2) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
      else v[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE != 1) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u]; }
      else acc += v[u];
    }
  }
  if (MODE == 1 && acc[0] == 12345.678f) sink[0] = acc[0];
}

template <class T, int U, int MODE, bool NT, bool CHUNKED>
double run(const void *a, void *b, size_t bytes, int grid, float *sink) {
  const size_t n = bytes / sizeof(T);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<T, U, MODE, NT, CHUNKED>), dim3(grid), dim3(256), 0, 0, (const T *)a, (T *)b, n, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return (MODE == 0 ? 2.0 : 1.0) * bytes / 1e9 / best;   // TB/s (GB per ms)
}

#define ROW(T, U, NT, CH)                                                                                                     \
  for (int grid : {1024, 2048, 4096, 16384}) {                                                                                \
    printf("%-7s U=%d nt=%d %-11s grid=%5d   copy %.2f   read %.2f   write %.2f TB/s\n", #T, U, NT, CH ? "chunked" : "grid-stride", \
           grid, run<T, U, 0, NT, CH>(a, b, bytes, grid, sink), run<T, U, 1, NT, CH>(a, b, bytes, grid, sink),                 \
           run<T, U, 2, NT, CH>(a, b, bytes, grid, sink));                                                                    \
    fflush(stdout);                                                                                                           \
  }

int main() {
  const size_t bytes = (size_t)2 << 30;
  void *a, *b; float *sink;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
  ROW(v4f, 1, false, false) ROW(v4f, 4, false, false) ROW(v4f, 8, false, false)
  ROW(v4f, 4, true, false) ROW(v4f, 4, false, true) ROW(v4f, 8, false, true) ROW(v4f, 4, true, true)
  ROW(v2f, 4, false, false) ROW(v2f, 8, false, false) ROW(v2f, 8, true, false) ROW(v2f, 8, false, true)
  return 0;
}
