// The atomic overlap-add SURVEY 7 (step 5) planned "for comparison" beside the segmented mixdown the library ships (VERDICT r05,
// "what's missing" 4; BASELINE.json north_star: "an atomic/segmented overlap-add into the per-capsule output timeline").
// Reference being replaced: generate_scene_audio_from_events, audiblelight/synthesize.py:373-378 (scene[:, a:b] += padded event audio).
//
//   segmented (what k_mixdown does): one workgroup per (capsule, 4096-sample tile); the events that overlap the tile are summed in
//                                    insertion order in registers, the tile is written once.  Deterministic, no atomics.
//   atomic:                          one workgroup per (event, capsule, 4096-sample chunk of the event); every sample is added to the
//                                    zeroed scene with a float atomicAdd.  The order in which events meet a sample is the hardware's.
// cfg2 shape: 64 events x 32 capsules x 192 000 samples into a (32, 2 880 000) scene; starts uniform over the scene.
// Reports: time of each, bytes / time, the largest difference between the two results, and whether the atomic result is the same
// bits run to run.  Build: hipcc --offload-arch=gfx950 -O3 atomic_mixdown_probe.hip -o atomic_mixdown_probe.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int E = 64, C = 32, LA = 192000, T = 2880000, TILE = 4096;

__global__ __launch_bounds__(256) void k_fill(float *x, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    x[i] = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
  }
}

// segmented: the events overlapping tile (c, t0) in insertion order (every event is tested: 64 compares per workgroup, as cheap as a table)
__global__ __launch_bounds__(256) void k_segmented(const float *__restrict__ x, const int *__restrict__ start, const float *__restrict__ scale,
                                                   float *__restrict__ scene) {
  const int c = blockIdx.y, t0 = blockIdx.x * TILE, tid = threadIdx.x;
  float4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = 0; e < E; ++e) {
    const int s = start[e];
    if (s + LA <= t0 || s >= t0 + TILE) continue;                  // workgroup-uniform
    const float g = scale[e];
    const float *row = x + ((size_t)e * C + c) * LA;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + 4 * (tid + 256 * j);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i = t + q - s; v[q] = (i >= 0 && i < LA && t + q < T) ? row[i] : 0.f; }
      acc[j].x += g * v[0]; acc[j].y += g * v[1]; acc[j].z += g * v[2]; acc[j].w += g * v[3];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = t0 + 4 * (tid + 256 * j);
    if (t + 3 < T) *reinterpret_cast<float4 *>(scene + (size_t)c * T + t) = acc[j];
    else for (int q = 0; q < 4 && t + q < T; ++q) scene[(size_t)c * T + t + q] = (&acc[j].x)[q];
  }
}

__global__ __launch_bounds__(256) void k_atomic(const float *__restrict__ x, const int *__restrict__ start, const float *__restrict__ scale,
                                                float *__restrict__ scene) {
  const int e = blockIdx.z, c = blockIdx.y, i0 = blockIdx.x * TILE, tid = threadIdx.x;
  const int s = start[e];
  const float g = scale[e];
  const float *row = x + ((size_t)e * C + c) * LA;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = i0 + tid + 256 * j;          // consecutive lanes hit consecutive addresses: one L2 atomic packet per 64 lanes
    if (i < LA && s + i < T) atomicAdd(scene + (size_t)c * T + s + i, g * row[i]);
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const size_t nx = (size_t)E * C * LA, ns = (size_t)C * T;
  float *x, *scene_a, *scene_b, *scene_c, *scale;
  int *start;
  CHECK(hipMalloc(&x, nx * 4)); CHECK(hipMalloc(&scene_a, ns * 4)); CHECK(hipMalloc(&scene_b, ns * 4)); CHECK(hipMalloc(&scene_c, ns * 4));
  CHECK(hipMalloc(&scale, E * 4)); CHECK(hipMalloc(&start, E * 4));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, nx, 12345u);
  std::vector<int> hs(E); std::vector<float> hg(E);
  srand(7);
  for (int e = 0; e < E; ++e) { hs[e] = rand() % (T - LA); hg[e] = 0.01f + 0.001f * e; }
  CHECK(hipMemcpy(start, hs.data(), E * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(scale, hg.data(), E * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%s; cfg2 mixdown: %d events x %d capsules x %d samples (%.2f GB) into (%d, %d) (%.2f GB)\n", prop.name, E, C, LA, nx * 4 / 1e9, C, T, ns * 4 / 1e9);
  float seg = 1e9f, atm = 1e9f, atm_with_zero = 1e9f;
  for (int rep = 0; rep < 8; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_segmented, dim3((T + TILE - 1) / TILE, C), dim3(256), 0, 0, x, start, scale, scene_a);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < seg) seg = ms;
    float *dst = (rep & 1) ? scene_c : scene_b;
    CHECK(hipEventRecord(e0));
    CHECK(hipMemsetAsync(dst, 0, ns * 4, 0));
    hipEvent_t em; CHECK(hipEventCreate(&em)); CHECK(hipEventRecord(em));
    hipLaunchKernelGGL(k_atomic, dim3((LA + TILE - 1) / TILE, C, E), dim3(256), 0, 0, x, start, scale, dst);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < atm_with_zero) atm_with_zero = ms;
    CHECK(hipEventElapsedTime(&ms, em, e1)); if (ms < atm) atm = ms;
    CHECK(hipEventDestroy(em));
  }
  std::vector<float> a(ns), b(ns), c(ns);
  CHECK(hipMemcpy(a.data(), scene_a, ns * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), scene_b, ns * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(c.data(), scene_c, ns * 4, hipMemcpyDeviceToHost));
  double worst = 0, peak = 0; size_t differ_runs = 0, differ_seg = 0;
  for (size_t i = 0; i < ns; ++i) {
    worst = fmax(worst, fabs((double)a[i] - b[i])); peak = fmax(peak, fabs((double)a[i]));
    differ_runs += memcmp(&b[i], &c[i], 4) != 0; differ_seg += memcmp(&a[i], &b[i], 4) != 0;
  }
  const double gb = (nx * 4 + ns * 4) / 1e9;
  printf("segmented (one write per tile, fixed order):  %.3f ms  (%.2f TB/s on x read + scene written)\n", seg, gb / seg);
  printf("atomic    (float atomicAdd per sample):       %.3f ms  (+ %.3f ms to zero the scene first = %.3f ms)  = %.2fx the segmented time\n",
         atm, atm_with_zero - atm, atm_with_zero, atm_with_zero / seg);
  printf("atomic vs segmented: max |difference| %.3g (peak %.3g: %.2g relative), %zu of %zu samples differ in their bits\n", worst, peak, worst / peak,
         differ_seg, ns);
  printf("atomic run to run:   %zu of %zu samples differ in their bits between two runs on the same inputs\n", differ_runs, ns);
  return 0;
}
