// Traffic model of cfg2's forward -> accumulate -> synthesis chain: the BYTES each stage moves, none of its arithmetic.
// Question ("move fewer bytes, or prove with a measurement that it cannot pay"): if the three stages ran as ONE
// persistent, event-pipelined kernel whose intermediate spectra (H, Y) live in rings small enough for the 256 MiB Infinity
// Cache, how much faster could the data move than through three launches over HBM-sized buffers?  This is the ceiling of
// that design: real kernels add FFT / FMA work on top and cannot beat it.
//
//   F(e,c,p): read one 32 KB IR partition, write one 64 KB H block            (k_forward_spectra_split)
//   M(e,c,t): read 12 x 4 KB (the bin tile of 12 H blocks), write 24 x 4 KB    (k_spectral_mac_static, per capsule)
//   S(e,c,k): read one 64 KB Y block, write 32 KB of event audio               (k_block_synthesis_split)
//
// mode A: three kernels, one workgroup per job, H (1.6 GB) and Y (3.2 GB) full size          = today's structure
// mode B: one persistent kernel; workgroups take tickets; ticket order = dependency order (F of unit u, M of unit u - a,
//         S of unit u - b per step; unit = (event, capsule)); H / Y rings of D units; per-unit completion counters with
//         agent-scope release / acquire; every spin is bounded (a stuck run reports instead of hanging).
// Build: hipcc --offload-arch=gfx950 -O3 pipeline_probe.hip -o pipeline_probe.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int E = 64, C = 32, P = 12, K = 24, TILES = 16;
constexpr int U = E * C;                       // units
constexpr int BLK4 = 4096;                     // float4 per 64 KB spectrum block
// tickets per unit and stage: by default one per job; -DCOARSE=1 hands out 1 F ticket (12 partitions), 4 M tickets (4 bin tiles
// each) and 4 S tickets (6 blocks each) per unit, so the ticket / flag traffic per byte drops 5-fold
#ifndef COARSE
#define COARSE 0
#endif
constexpr int NF = COARSE ? 1 : P, NM = COARSE ? 4 : TILES, NS = COARSE ? 4 : K, PER_STEP = NF + NM + NS;
constexpr int GF = P / NF, GM = TILES / NM, GS = K / NS;   // jobs per ticket

#ifndef NT_STORES
#define NT_STORES 0
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(float4 *p, float4 v) {
#if NT_STORES
  v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p));
#else
  *p = v;
#endif
}

// ring accesses: MODE 0 = plain (cached, needs agent-scope release / acquire fences around the hand-over),
// MODE 1 = system-coherent instructions (sc0 sc1: written through, read past the non-coherent caches), no agent fences at all
// The compiler tracks these (it places the s_waitcnt itself), unlike inline-asm loads.  aux bits on gfx94x/95x: 1 = sc0, 16 = sc1.
typedef int v4i __attribute__((ext_vector_type(4)));
// (uniform base, per-lane float4 index): MODE 1 goes through a buffer descriptor of the base with the lane's byte offset
template <int MODE>
__device__ __forceinline__ float4 ring_ld(const float4 *base, size_t idx) {
  if constexpr (MODE == 1) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(base), 0, 0xFFFFFFF0u, 0x00020000);
    const v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(unsigned)(idx * 16), 0, 17);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
  } else {
    return base[idx];
  }
}
template <int MODE>
__device__ __forceinline__ void wait_loads() {}
template <int MODE>
__device__ __forceinline__ void ring_st(float4 *base, size_t idx, float4 v) {
  if constexpr (MODE == 1) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFF0u, 0x00020000);
    const v4i t = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)(unsigned)(idx * 16), 0, 17);
  } else {
    st(base + idx, v);
  }
}

struct Bufs {
  const float4 *ir;   // U * P * 2048 float4 (32 KB per partition)
  float4 *h;          // ring: Dh units * P blocks
  float4 *y;          // ring: Dy units * K blocks
  float4 *x;          // U * K * 2048 float4
  int dh, dy;
};

template <int MODE>
__device__ __forceinline__ void job_f(const Bufs &b, int u, int p, int tid) {
  const float4 *src = b.ir + ((size_t)u * P + p) * 2048;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float4 v = src[tid + 256 * j]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  const size_t dst = ((size_t)(u % b.dh) * P + p) * BLK4;
#pragma unroll
  for (int j = 0; j < 16; ++j) ring_st<MODE>(b.h, dst + tid + 256 * j, make_float4(acc.x + j, acc.y, acc.z, acc.w));
}

template <int MODE>
__device__ __forceinline__ void job_m(const Bufs &b, int u, int t, int tid) {
  const size_t src = (size_t)(u % b.dh) * P * BLK4 + t * 256 + tid;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), v[P];
#pragma unroll
  for (int p = 0; p < P; ++p) v[p] = ring_ld<MODE>(b.h, src + (size_t)p * BLK4);
  wait_loads<MODE>();
#pragma unroll
  for (int p = 0; p < P; ++p) { acc.x += v[p].x; acc.y += v[p].y; acc.z += v[p].z; acc.w += v[p].w; }
  const size_t dst = (size_t)(u % b.dy) * K * BLK4 + t * 256 + tid;
#pragma unroll
  for (int k = 0; k < K; ++k) ring_st<MODE>(b.y, dst + (size_t)k * BLK4, make_float4(acc.x + k, acc.y, acc.z, acc.w));
}

template <int MODE>
__device__ __forceinline__ void job_s(const Bufs &b, int u, int k, int tid) {
  const size_t src = ((size_t)(u % b.dy) * K + k) * BLK4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = ring_ld<MODE>(b.y, src + tid + 256 * j);
  wait_loads<MODE>();
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
  float4 *dst = b.x + ((size_t)u * K + k) * 2048;
#pragma unroll
  for (int j = 0; j < 8; ++j) st(dst + tid + 256 * j, make_float4(acc.x + j, acc.y, acc.z, acc.w));
}

// ---------------------------------------------------------------- mode A: one launch per stage, one workgroup per job
__global__ __launch_bounds__(256) void k_f(Bufs b) { job_f<0>(b, blockIdx.x / P, blockIdx.x % P, threadIdx.x); }
__global__ __launch_bounds__(256) void k_m(Bufs b) { job_m<0>(b, blockIdx.x / TILES, blockIdx.x % TILES, threadIdx.x); }
__global__ __launch_bounds__(256) void k_s(Bufs b) { job_s<0>(b, blockIdx.x / K, blockIdx.x % K, threadIdx.x); }
// the accumulate as the library runs it: one workgroup per (event, bin tile) looping over the capsules
__global__ __launch_bounds__(256) void k_m_loop(Bufs b) {
  const int e = blockIdx.x / TILES, t = blockIdx.x % TILES;
  for (int c = 0; c < C; ++c) job_m<0>(b, e * C + c, t, threadIdx.x);
}

// ---------------------------------------------------------------- mode B: persistent, ticketed, rings
struct Sync {
  int *ticket;        // 1
  int *done_f, *done_m, *done_s;   // U each
  int *stuck;         // 1: number of waits that ran out of patience
  long long *spins;   // 1: total spin iterations (how much the pipeline waited)
};

__device__ __forceinline__ bool wait_for(const int *counter, int want, Sync s) {
  int n = 0;
  while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {   // relaxed: an acquire here would invalidate the caches on every poll
    __builtin_amdgcn_s_sleep(8);
    ++n;
    if ((n & 1023) == 0 && __hip_atomic_load(s.stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;   // somebody gave up: everybody does
    if (n > 200000) { if (threadIdx.x == 0) atomicAdd(s.stuck, 1); return false; }   // about 40 ms: a hung pipeline reports, it does not hang
  }
  if (n && threadIdx.x == 0) atomicAdd((unsigned long long *)s.spins, (unsigned long long)n);
  return true;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_pipeline(Bufs b, Sync s, int lag_m, int lag_s, int n_steps) {
  __shared__ int sh_ticket;
  const int tid = threadIdx.x;
  const int total = n_steps * PER_STEP;
  while (true) {
    if (tid == 0) sh_ticket = atomicAdd(s.ticket, 1);
    __syncthreads();
    const int ticket = sh_ticket;
    __syncthreads();
    if (ticket >= total) return;
    const int step = ticket / PER_STEP, j = ticket % PER_STEP;
    int *done = nullptr;
    if (j < NF) {
      const int u = step;
      if (u >= U) continue;
      if (u >= b.dh) wait_for(s.done_m + (u - b.dh), NM, s);          // the H ring slot has been consumed
      for (int g = 0; g < GF; ++g) job_f<MODE>(b, u, j * GF + g, tid);
      done = s.done_f + u;
    } else if (j < NF + NM) {
      const int u = step - lag_m;
      if (u < 0 || u >= U) continue;
      wait_for(s.done_f + u, NF, s);
      if (u >= b.dy) wait_for(s.done_s + (u - b.dy), NS, s);          // the Y ring slot has been consumed
      if constexpr (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int g = 0; g < GM; ++g) job_m<MODE>(b, u, (j - NF) * GM + g, tid);
      done = s.done_m + u;
    } else {
      const int u = step - lag_s;
      if (u < 0 || u >= U) continue;
      wait_for(s.done_m + u, NM, s);
      if constexpr (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int g = 0; g < GS; ++g) job_s<MODE>(b, u, (j - NF - NM) * GS + g, tid);
      done = s.done_s + u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's stores are complete (MODE 1: written through, acknowledged)
    __syncthreads();
    if (tid == 0) {
      if constexpr (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // ONE L2 write-back per job, not one per thread
      __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main(int argc, char **argv) {
  int wg_per_cu = argc > 1 ? atoi(argv[1]) : 4;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("%s, %d CUs; E=%d C=%d P=%d K=%d; NT_STORES=%d COARSE=%d (tickets per unit: %d F, %d M, %d S)\n", prop.name, n_cu, E, C, P, K,
         NT_STORES, COARSE, NF, NM, NS);
  const size_t ir_bytes = (size_t)U * P * 32768, x_bytes = (size_t)U * K * 32768;
  const size_t h_full = (size_t)U * P * 65536, y_full = (size_t)U * K * 65536;
  float4 *ir, *h, *y, *x;
  CHECK(hipMalloc(&ir, ir_bytes)); CHECK(hipMalloc(&h, h_full)); CHECK(hipMalloc(&y, y_full)); CHECK(hipMalloc(&x, x_bytes));
  CHECK(hipMemset(ir, 0, ir_bytes));
  CHECK(hipMemset(h, 0, h_full)); CHECK(hipMemset(y, 0, y_full)); CHECK(hipMemset(x, 0, x_bytes));
  Sync s;
  int *ints; long long *spins;
  CHECK(hipMalloc(&ints, sizeof(int) * (2 + 3 * U))); CHECK(hipMalloc(&spins, 8));
  s.ticket = ints; s.stuck = ints + 1; s.done_f = ints + 2; s.done_m = s.done_f + U; s.done_s = s.done_m + U; s.spins = spins;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double gb_f = (ir_bytes + h_full) / 1e9, gb_m = (h_full + y_full) / 1e9, gb_s = (y_full + x_bytes) / 1e9;
  const double gb_all = gb_f + gb_m + gb_s;
  printf("bytes per pass: F %.2f GB, M %.2f GB, S %.2f GB, total %.2f GB\n", gb_f, gb_m, gb_s, gb_all);

  {  // ---- mode A
    Bufs b{ir, h, y, x, U, U};
    float best[4] = {1e9f, 1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 6; ++rep) {
      float ms[4];
      hipEvent_t ev[5]; for (auto &e : ev) CHECK(hipEventCreate(&e));
      CHECK(hipEventRecord(ev[0]));
      hipLaunchKernelGGL(k_f, dim3(U * P), dim3(256), 0, 0, b);
      CHECK(hipEventRecord(ev[1]));
      hipLaunchKernelGGL(k_m, dim3(U * TILES), dim3(256), 0, 0, b);
      CHECK(hipEventRecord(ev[2]));
      hipLaunchKernelGGL(k_s, dim3(U * K), dim3(256), 0, 0, b);
      CHECK(hipEventRecord(ev[3]));
      hipLaunchKernelGGL(k_m_loop, dim3(E * TILES), dim3(256), 0, 0, b);
      CHECK(hipEventRecord(ev[4]));
      CHECK(hipEventSynchronize(ev[4]));
      for (int i = 0; i < 4; ++i) { CHECK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1])); if (ms[i] < best[i]) best[i] = ms[i]; }
      for (auto &e : ev) CHECK(hipEventDestroy(e));
    }
    printf("mode A (three launches, HBM-sized H / Y): F %.3f ms (%.2f TB/s)  M %.3f ms (%.2f TB/s)  S %.3f ms (%.2f TB/s)  sum %.3f ms (%.2f TB/s)"
           "   [M as a capsule loop: %.3f ms]\n", best[0], gb_f / best[0], best[1], gb_m / best[1], best[2], gb_s / best[2],
           best[0] + best[1] + best[2], gb_all / (best[0] + best[1] + best[2]), best[3]);
  }

  // ---- mode B: ring depth D (units of 2.3 MB: 0.77 MB H + 1.5 MB Y), lags in units
  struct Cfg { int d, lag_m, lag_s; };
  const Cfg cfgs[] = {{U, 24, 48}, {U, 64, 128}, {32, 8, 16}, {64, 16, 32}, {96, 24, 48}, {64, 24, 48}, {128, 48, 96}, {128, 64, 128}};
  // ring memory kinds: 0 ordinary hipMalloc (coarse-grained), 1 fine-grained, 2 uncached
  float4 *hk[3] = {h, nullptr, nullptr}, *yk[3] = {y, nullptr, nullptr};
  const size_t ring_max = 128;
  for (int kind = 1; kind < 3; ++kind) {
    const unsigned flag = kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached;
    if (hipExtMallocWithFlags((void **)&hk[kind], ring_max * P * 65536, flag) != hipSuccess ||
        hipExtMallocWithFlags((void **)&yk[kind], ring_max * K * 65536, flag) != hipSuccess) { hk[kind] = yk[kind] = nullptr; (void)hipGetLastError(); }
  }
  const char *kind_name[3] = {"coarse", "fine-grained", "uncached"};
  for (int variant = 0; variant < 4; ++variant) {
    const int mode = variant < 1 ? 0 : 1, kind = 0;    // mode 0: cached accesses + agent fences; 1: sc0 sc1 accesses, no agent fences
    const int wpc = variant == 2 ? wg_per_cu / 2 : variant == 3 ? wg_per_cu * 2 : wg_per_cu;
    if (!hk[kind]) { printf("memory kind %s not available\n", kind_name[kind]); continue; }
    for (const Cfg &c : cfgs) {
      if (c.lag_s >= c.d + c.lag_m || c.lag_m >= c.d) continue;   // ticket order must be dependency order
      if (kind != 0 && c.d > (int)ring_max) continue;
      Bufs b{ir, hk[kind], yk[kind], x, c.d, c.d};
      float best = 1e9f; int stuck = 0; long long spin = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemsetAsync(ints, 0, sizeof(int) * (2 + 3 * U), 0)); CHECK(hipMemsetAsync(spins, 0, 8, 0));
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_pipeline<0>, dim3(n_cu * wpc), dim3(256), 0, 0, b, s, c.lag_m, c.lag_s, U + c.lag_s);
        else hipLaunchKernelGGL(k_pipeline<1>, dim3(n_cu * wpc), dim3(256), 0, 0, b, s, c.lag_m, c.lag_s, U + c.lag_s);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CHECK(hipMemcpy(&stuck, s.stuck, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&spin, spins, 8, hipMemcpyDeviceToHost));
        if (stuck) break;
      }
      printf("mode B  %s  %-12s %d wg/CU  ring %4d units (%6.1f MB)  lags M %2d S %2d: %.3f ms (%.2f TB/s)  spins %lld%s\n",
             mode ? "sc0sc1 no-fence" : "cached + fences", kind_name[kind], wpc, c.d,
             c.d * (P + K) * 65536.0 / 1e6, c.lag_m, c.lag_s, best, gb_all / best, spin, stuck ? "  STUCK (bounded wait ran out)" : "");
      fflush(stdout);
    }
  }
  return 0;
}
