// k_block_synthesis's access pattern without its arithmetic: per workgroup (256 threads) one 64 KB spectrum read as
// (k, M-k) pairs [mode 0] or front to back [mode 1], 32 KB of samples written.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k_pattern(const float2 *__restrict__ y, float2 *__restrict__ out, int M) {
  const int tid = threadIdx.x, T = 256;
  const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const float2 *in = y + blk * M;
  float2 a[16], b[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int k = tid + T * m;
    a[m] = in[k];
    b[m] = MODE == 0 ? in[k == 0 ? M / 2 : M - k] : in[M / 2 + k];
  }
  float2 *o = out + blk * (M / 2) + tid;
#pragma unroll
  for (int m = 0; m < 16; ++m) o[T * m] = make_float2(a[m].x + b[m].y, a[m].y - b[m].x);
}

template <int MODE>
void run(const char *name) {
  const int E = 64, C = 32, K = 24, M = 8192;
  float2 *y, *out;
  const size_t yb = (size_t)E * C * K * M * 8, ob = yb / 2;
  (void)hipMalloc(&y, yb); (void)hipMalloc(&out, ob); (void)hipMemset(y, 0, yb);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_pattern<MODE>), dim3(K, C, E), dim3(256), 0, 0, y, out, M);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms, %.2f GB -> %.2f TB/s\n", name, ms, (yb + ob) / 1e9, (yb + ob) / 1e9 / ms);
  (void)hipFree(y); (void)hipFree(out);
}

int main() {
  run<0>("(k, M-k) pairs");
  run<1>("front to back");
  return 0;
}
