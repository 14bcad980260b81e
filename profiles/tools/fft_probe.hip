// How fast is the in-LDS transform alone (no HBM traffic)?  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <al_fft.h>

template <int LOG2M>
__global__ __launch_bounds__(al::fft_threads(LOG2M)) void k_fft_only(const float2 *tw, float2 *sink, int iters) {
  __shared__ float2 s[al::fft_lds_elems(LOG2M)];
  constexpr int T = (1 << LOG2M) / 16;
  const int tid = threadIdx.x;
  float2 v[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = make_float2((float)(tid + m) * 1e-3f, (float)(blockIdx.x & 7));
  for (int it = 0; it < iters; ++it) {
    al::fft_regs_to_regs<LOG2M, -1>(v, s, tw, tid);
#pragma unroll
    for (int m = 0; m < 16; ++m) { v[m].x *= 1e-2f; v[m].y *= 1e-2f; }
    __syncthreads();
  }
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int m = 0; m < 16; ++m) { acc.x += v[m].x; acc.y += v[m].y; }
  if (acc.x == 12345.678f) sink[blockIdx.x * T + tid] = acc;
}

__global__ void k_tw(float2 *tw, int m) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) { double s, c; sincospi(-(double)k / m, &s, &c); tw[k] = make_float2((float)c, (float)s); }
}

template <int LOG2M>
void run(float2 *tw, float2 *sink, int wgs_per_cu) {
  const int M = 1 << LOG2M, iters = 200, grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(k_tw, dim3((M + 255) / 256), dim3(256), 0, 0, tw, M);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_fft_only<LOG2M>), dim3(grid), dim3(al::fft_threads(LOG2M)), 0, 0, tw, sink, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_fft_only<LOG2M>), dim3(grid), dim3(al::fft_threads(LOG2M)), 0, 0, tw, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per_cu_us = ms * 1e3 / (double)(iters * wgs_per_cu);
  printf("M=%5d  %d WG/CU launched  %.3f ms  -> %.2f us per transform per CU (%.2f us per 8192 points)\n", M, wgs_per_cu, ms,
         per_cu_us, per_cu_us * 8192.0 / M);
}

// variant: 16 KB of dynamic LDS padding to force fewer resident workgroups
template <int LOG2M>
__global__ __launch_bounds__(al::fft_threads(LOG2M)) void k_fft_lowocc(const float2 *tw, float2 *sink, int iters) {
  __shared__ float2 s[al::fft_lds_elems(LOG2M) * 2];   // doubles the LDS footprint: halves the workgroups per CU
  const int tid = threadIdx.x;
  float2 v[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = make_float2((float)(tid + m) * 1e-3f, (float)(blockIdx.x & 7));
  for (int it = 0; it < iters; ++it) {
    al::fft_regs_to_regs<LOG2M, -1>(v, s, tw, tid);
#pragma unroll
    for (int m = 0; m < 16; ++m) { v[m].x *= 1e-2f; v[m].y *= 1e-2f; }
    __syncthreads();
  }
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int m = 0; m < 16; ++m) { acc.x += v[m].x; acc.y += v[m].y; }
  if (acc.x == 12345.678f) sink[blockIdx.x * 64 + tid] = acc;
}

template <int LOG2M>
void run_low(float2 *tw, float2 *sink, int wgs_per_cu) {
  const int M = 1 << LOG2M, iters = 200, grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(k_tw, dim3((M + 255) / 256), dim3(256), 0, 0, tw, M);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_fft_lowocc<LOG2M>), dim3(grid), dim3(al::fft_threads(LOG2M)), 0, 0, tw, sink, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_fft_lowocc<LOG2M>), dim3(grid), dim3(al::fft_threads(LOG2M)), 0, 0, tw, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per_cu_us = ms * 1e3 / (double)(iters * wgs_per_cu);
  printf("LOWOCC M=%5d  %d WG/CU launched  %.3f ms  -> %.2f us per transform per CU (%.2f us per 8192 points)\n", M, wgs_per_cu, ms,
         per_cu_us, per_cu_us * 8192.0 / M);
}

int main() {
  float2 *tw, *sink;
  hipMalloc(&tw, 1 << 20); hipMalloc(&sink, 64 << 20);
  run<12>(tw, sink, 4); run<12>(tw, sink, 8);
  run<13>(tw, sink, 2); run<13>(tw, sink, 4);
  run<14>(tw, sink, 1); run<11>(tw, sink, 8); run<10>(tw, sink, 8);
  run_low<12>(tw, sink, 2); run_low<12>(tw, sink, 4); run_low<11>(tw, sink, 4);
  return 0;
}
