// Streaming ceilings on this MI355X for the access mixes of the synthesis kernels (measurement tool only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int RD, int WR, typename T>  // each thread-iteration reads RD vectors and writes WR vectors
__global__ __launch_bounds__(256) void k_stream(const T *__restrict__ in, T *__restrict__ out, size_t n_iter) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_iter; i += (size_t)gridDim.x * 256) {
    T acc = in[i];
#pragma unroll
    for (int r = 1; r < RD; ++r) {
      T v = in[i + r * n_iter];
      acc.x += v.x;
      acc.y += v.y;
    }
#pragma unroll
    for (int w = 0; w < WR; ++w) out[i + w * n_iter] = acc;
  }
}

template <int RD, int WR, typename T>
void run(const char *name, void *a, void *b, size_t bytes_total, int grid) {
  size_t n_iter = bytes_total / sizeof(T) / (RD + WR);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_stream<RD, WR, T>), dim3(grid), dim3(256), 0, 0, (const T *)a, (T *)b, n_iter);
  hipEventRecord(e0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_stream<RD, WR, T>), dim3(grid), dim3(256), 0, 0, (const T *)a, (T *)b, n_iter);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double gb = (double)n_iter * sizeof(T) * (RD + WR) / 1e9;
  printf("%-28s grid=%6d  %.3f ms  %.0f GB/s (rd %.2f GB wr %.2f GB)\n", name, grid, ms / reps, gb / (ms / reps * 1e-3),
         (double)n_iter * sizeof(T) * RD / 1e9, (double)n_iter * sizeof(T) * WR / 1e9);
}

int main() {
  const size_t bytes = (size_t)6 << 30;
  void *a, *b;
  hipMalloc(&a, bytes);
  hipMalloc(&b, bytes);
  hipMemset(a, 0, bytes);
  hipMemset(b, 0, bytes);
  for (int grid : {2048, 8192, 65536}) {
    run<1, 1, float4>("copy float4 1r:1w", a, b, (size_t)4 << 30, grid);
    run<1, 1, float2>("copy float2 1r:1w", a, b, (size_t)4 << 30, grid);
    run<1, 2, float4>("float4 1r:2w (mac-like)", a, b, (size_t)4608 << 20, grid);
    run<1, 2, float2>("float2 1r:2w (mac-like)", a, b, (size_t)4608 << 20, grid);
    run<2, 1, float2>("float2 2r:1w (synth-like)", a, b, (size_t)4608 << 20, grid);
    run<2, 1, float4>("float4 2r:1w (synth-like)", a, b, (size_t)4608 << 20, grid);
    run<4, 1, float4>("float4 4r:1w (mix-like)", a, b, (size_t)2 << 30, grid);
  }
  return 0;
}
