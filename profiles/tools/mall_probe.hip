// Does a buffer that is written by one kernel and read by the next stay in the 256 MiB Infinity Cache?
// write kernel + read kernel over the same buffer, repeated; sizes from 32 MB to 4 GB.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_write(float4 *p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_read(const float4 *p, size_t n, float *sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 x = p[i]; acc += x.x + x.y + x.z + x.w; }
  if (acc == 12345.678f) sink[0] = acc;
}
// one kernel: every workgroup writes its own 64 KB piece, then reads the piece of its neighbour (id + 8: same XCD) written
// in the PREVIOUS iteration (no synchronisation needed for a timing probe), 50 iterations over the same buffer
__global__ __launch_bounds__(256) void k_pingpong(float4 *p, int iters, float *sink) {
  const size_t piece = 4096;  // float4 per workgroup = 64 KB
  float4 *mine = p + (size_t)blockIdx.x * piece;
  const float4 *other = p + (size_t)((blockIdx.x + 8) % gridDim.x) * piece;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < (int)piece; i += 256) mine[i] = make_float4(acc, it, i, 1.f);
    for (int i = threadIdx.x; i < (int)piece; i += 256) { float4 x = other[i]; acc += x.x + x.w; }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  float *sink; (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (size_t mb : {32, 64, 96, 128, 192, 256, 512, 1024, 4096}) {
    const size_t n = mb * 1024 * 1024 / 16;
    float4 *p; (void)hipMalloc(&p, n * 16);
    const int reps = mb <= 256 ? 40 : 8;
    for (int warm = 0; warm < 2; ++warm) {
      (void)hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, p, n, (float)r);
        hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, p, n, sink);
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("write kernel + read kernel over %4zu MB: %.3f ms per pair -> %.2f TB/s (bytes written + read)\n", mb, ms / reps,
           2.0 * mb * 1.048576e-3 / (ms / reps) );
    (void)hipFree(p);
  }
  {
    const int wgs = 2048, iters = 50;  // 2048 x 64 KB = 128 MB, persistent-style: 8 workgroups per CU slot... 2048 > resident 
    float4 *p; (void)hipMalloc(&p, (size_t)wgs * 65536);
    for (int warm = 0; warm < 2; ++warm) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_pingpong, dim3(wgs), dim3(256), 0, 0, p, iters, sink);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("in-kernel write-own / read-neighbour, 128 MB buffer, %d iterations: %.3f ms -> %.2f TB/s\n", iters, ms,
           2.0 * wgs * 65536.0 * iters / 1e9 / ms);
    (void)hipFree(p);
  }
  return 0;
}
