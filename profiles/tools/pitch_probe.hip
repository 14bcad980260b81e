// Does the 64 KB pitch between the block spectra cost HBM bandwidth (channel camping)?  The spectral MAC's access
// pattern without its arithmetic: every workgroup reads 12 rows of 4 KB at row pitch `pitch` and writes 12 rows.
// Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_pattern(const float4 *__restrict__ h, float4 *__restrict__ y, int pitch4, int rows_per_item,
                                                 int k_rows) {
  // item = (e, c) = blockIdx.z, blockIdx.y / 2; k-tile = blockIdx.y % 2; slice = blockIdx.x (256 threads x 16 B = 4 KB)
  const int64_t item = (int64_t)blockIdx.z * (gridDim.y / 2) + blockIdx.y / 2;
  const int tile = blockIdx.y & 1;
  const float4 *hp = h + (item * rows_per_item) * pitch4 + blockIdx.x * 256 + threadIdx.x;
  float4 acc[12];
#pragma unroll
  for (int p = 0; p < 12; ++p) acc[p] = hp[(int64_t)p * pitch4];
  float4 *yp = y + (item * k_rows + tile * 12) * pitch4 + blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    float4 v = acc[k];
    v.x += acc[(k + 1) % 12].y;
    yp[(int64_t)k * pitch4] = v;
  }
}

int main() {
  const int E = 64, C = 32, P = 12, K = 24, M = 8192;  // complex points per row = M -> M/2 float4
  for (int pad : {0, 16, 32, 64, 128, 272}) {
    const int pitch4 = (M + pad) / 2;
    float4 *h, *y;
    const size_t hb = (size_t)E * C * P * pitch4 * 16, yb = (size_t)E * C * K * pitch4 * 16;
    (void)hipMalloc(&h, hb); (void)hipMalloc(&y, yb);
    (void)hipMemset(h, 0, hb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    dim3 grid(M / 2 / 256, C * 2, E);
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_pattern, grid, dim3(256), 0, 0, h, y, pitch4, P, K);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // bytes: H read twice (two k-tiles; the second may hit L2) counted once + Y written once
    const double gb = ((double)E * C * P * M * 8 + (double)E * C * K * M * 8) / 1e9;
    printf("row pitch %d complex (+%d): %.3f ms, %.2f GB designed -> %.2f TB/s\n", M + pad, pad, ms, gb, gb / ms);
    (void)hipFree(h); (void)hipFree(y);
  }
  return 0;
}
