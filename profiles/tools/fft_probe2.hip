// What does the in-LDS transform wait on?  Same kernel as fft_probe.hip, built in several modes:
//   -DPROBE_MODE=0 baseline   1: workgroup barriers removed (results wrong, timing only)
// and run at 2 and 1 resident workgroups per CU.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#if PROBE_MODE == 1
#define __syncthreads() __builtin_amdgcn_wave_barrier()
#endif
#include <al_fft.h>

template <int LOG2M, int E, int PADK>
__global__ __launch_bounds__((al::FftGeom<LOG2M, E>::T)) void k_fft_only(const float2 *tw, float2 *sink, int iters) {
  using G = al::FftGeom<LOG2M, E>;
  __shared__ float2 s[G::LDS_ELEMS + PADK * 1024 / 8];
  constexpr int T = G::T;
  const int tid = threadIdx.x;
  float2 v[E];
#pragma unroll
  for (int m = 0; m < E; ++m) v[m] = make_float2((float)(tid + m) * 1e-3f, (float)(blockIdx.x & 7));
  for (int it = 0; it < iters; ++it) {
    al::fft_regs_to_regs<G, -1>(v, s, tw, tid);
#pragma unroll
    for (int m = 0; m < E; ++m) { v[m].x *= 1e-2f; v[m].y *= 1e-2f; }
    __syncthreads();
  }
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int m = 0; m < E; ++m) { acc.x += v[m].x; acc.y += v[m].y; }
  if (acc.x == 12345.678f) sink[blockIdx.x * T + tid] = acc;
  if (PADK && acc.y == 12345.678f) sink[0] = s[G::LDS_ELEMS + tid];
}

__global__ void k_tw(float2 *tw, int m) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) { double s, c; sincospi(-(double)k / m, &s, &c); tw[k] = make_float2((float)c, (float)s); }
}

template <int LOG2M, int E, int PADK>
void run(float2 *tw, float2 *sink, int wgs_per_cu, const char *note) {
  const int M = 1 << LOG2M, iters = 200, grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL(k_tw, dim3((M + 255) / 256), dim3(256), 0, 0, tw, M);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_fft_only<LOG2M, E, PADK>), dim3(grid), dim3(al::FftGeom<LOG2M, E>::T), 0, 0, tw, sink, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_fft_only<LOG2M, E, PADK>), dim3(grid), dim3(al::FftGeom<LOG2M, E>::T), 0, 0, tw, sink, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double per_cu_us = ms * 1e3 / (double)(iters * wgs_per_cu);
  printf("mode %d  M=%5d E=%d  %-22s %.3f ms -> %.2f us per transform per CU (%.2f us per 8192 points)\n", PROBE_MODE, M, E, note, ms,
         per_cu_us, per_cu_us * 8192.0 / M);
}

int main() {
  float2 *tw, *sink;
  (void)hipMalloc(&tw, 1 << 20); (void)hipMalloc(&sink, 64 << 20);
  run<13, 16, 0>(tw, sink, 2, "2 WG/CU resident");
  run<13, 32, 0>(tw, sink, 2, "2 WG/CU resident");
  run<13, 32, 24>(tw, sink, 1, "1 WG/CU resident");
  run<14, 16, 0>(tw, sink, 1, "1 WG/CU resident");
  run<14, 32, 0>(tw, sink, 1, "1 WG/CU resident");
  run<12, 16, 0>(tw, sink, 4, "4 WG/CU resident");
  run<12, 32, 0>(tw, sink, 4, "4 WG/CU resident");
  run<11, 16, 0>(tw, sink, 8, "8 WG/CU resident");
  return 0;
}
