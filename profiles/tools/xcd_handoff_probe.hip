// Traffic model of the accumulate -> synthesis hand-off of Y INSIDE ONE XCD (round 6; VERDICT r05 "next round" item 2).
// Round 3's pipeline_probe.hip handed H and Y between roles chip-wide (rings sized for the 256 MiB Infinity Cache, sc0 sc1
// accesses because the XCDs' L2s are not coherent with each other): ceiling -0.47 ms per cfg2 scene with no arithmetic.  This
// probe asks the question that one left open: if the producers (accumulate: one workgroup per (event, bin tile) looping over the
// capsules, as k_spectral_mac_static does) and the consumers (inverse transforms of the blocks they publish) of an event run on
// the SAME XCD, so that the hand-off needs no cross-XCD coherence and the Y ring can be small enough for that XCD's 4 MB L2,
// how fast can the bytes of the two stages move?  No arithmetic: this is the design's ceiling.
//
//   producer (e, tile):  for c in capsules: read P x 4 KB of H (HBM), wait for ring slot c % R, write K x 4 KB of Y into it
//   consumer (e, c, k2): wait for the 16 producers of (e, c); read two 64 KB Y blocks (L2, sc1 loads: bypass this CU's L1);
//                        release the slot as soon as the blocks are in registers; write 2 x 32 KB of event audio (HBM)
//   workgroups are 512 threads (the accumulate's two k-tiles; two blocks per consumer), persistent, and take tickets from the
//   queue of THEIR XCD (HW_REG_XCC_ID); ticket order = dependency order, every wait is bounded.
//
//   mode A = two launches over an HBM-sized Y (what the library does): the same jobs, one workgroup per job.
// Build: hipcc --offload-arch=gfx950 -O3 xcd_handoff_probe.hip -o xcd_handoff_probe.  Measurement tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef CFG_E
#define CFG_E 32   /* cfg4: 32 events x 32 capsules, P = 6, K = 24 */
#define CFG_C 32
#define CFG_P 6
#endif
constexpr int E = CFG_E, C = CFG_C, P = CFG_P, K = 24, TILES = 16, NXCD = 8;
constexpr int BLK4 = 4096;                     // float4 per 64 KB spectrum block
constexpr int KT = K / 2;                      // blocks per k-tile (one 256-thread half each)
constexpr int CONS_PER_STEP = K / 2;           // consumer jobs per (event, capsule): two blocks each
constexpr int TICKETS_PER_EVENT = TILES + C * CONS_PER_STEP;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_nt(float4 *p, float4 v) {
  v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p));
}
// sc1 load (aux 16): served by the L2, never by this CU's L1
__device__ __forceinline__ float4 ld_sc1(const float4 *base, size_t idx) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(base), 0, 0xFFFFFFF0u, 0x00020000);
  const v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(unsigned)(idx * 16), 0, 16);
  return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

struct Bufs {
  const float4 *h;    // [E][C][P][BLK4]
  float4 *y;          // mode A: [E][C][K][BLK4]; mode X: [NXCD][2][R][K][BLK4]
  float4 *x;          // [E][C][K][2048]
  int r;              // ring slots (capsule steps)
  int par;            // events in flight per XCD (their producers take tickets together), each with a ring of its own
  int sleep;          // s_sleep argument of the polls (x 64 cycles)
};
struct Sync {
  int *ticket;        // [NXCD * 32]  (one 128-B line each)
  int *full;          // [E * C]   producers that have published (e, c)
  int *taken;         // [E * C]   consumer jobs that have loaded their blocks of (e, c)
  int *stuck;
  long long *spins;   // [2]: producer spins, consumer spins
  int *census;        // [NXCD] workgroups seen per XCD
};

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}

__device__ __forceinline__ bool wait_for(const int *counter, int want, Sync s, int which, int sleep) {
  int n = 0;
  while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    for (int i = 0; i < sleep; ++i) __builtin_amdgcn_s_sleep(1);
    ++n;
    if ((n & 1023) == 0 && __hip_atomic_load(s.stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    if (n > 400000 / (sleep > 0 ? sleep : 1) + 1000) { atomicAdd(s.stuck, 1); return false; }
  }
  if (n) atomicAdd((unsigned long long *)s.spins + which, (unsigned long long)n);
  return true;
}

// ---------------------------------------------------------------- mode A
__global__ __launch_bounds__(512) void k_m_loop(Bufs b) {      // one workgroup per (event, bin tile), both k-tiles, capsule loop
  const int e = blockIdx.x / TILES, t = blockIdx.x % TILES, tid = threadIdx.x & 255, sub = threadIdx.x >> 8;
  for (int c = 0; c < C; ++c) {
    const float4 *src = b.h + ((size_t)(e * C + c) * P) * BLK4 + t * 256 + tid;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < P; ++p) { const float4 v = src[(size_t)p * BLK4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    float4 *dst = b.y + ((size_t)(e * C + c) * K + sub * KT) * BLK4 + t * 256 + tid;
#pragma unroll
    for (int k = 0; k < KT; ++k) st_nt(dst + (size_t)k * BLK4, make_float4(acc.x + k, acc.y, acc.z, acc.w));
    __syncthreads();
  }
}
__global__ __launch_bounds__(512) void k_s(Bufs b) {           // one workgroup per two blocks
  const int u = blockIdx.x / CONS_PER_STEP, k = (blockIdx.x % CONS_PER_STEP) * 2 + (threadIdx.x >> 8), tid = threadIdx.x & 255;
  const float4 *src = b.y + ((size_t)u * K + k) * BLK4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = src[tid + 256 * j];
#pragma unroll
  for (int j = 0; j < 16; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
  float4 *dst = b.x + ((size_t)u * K + k) * 2048;
#pragma unroll
  for (int j = 0; j < 8; ++j) st_nt(dst + tid + 256 * j, make_float4(acc.x + j, acc.y, acc.z, acc.w));
}

// ---------------------------------------------------------------- mode X: persistent, per-XCD queues, L2-sized ring
template <bool PLAIN_LOADS>
__global__ __launch_bounds__(512) void k_xcd(Bufs b, Sync s) {
  __shared__ int sh_ticket;
  const int tid = threadIdx.x & 255, sub = threadIdx.x >> 8;
  const int xcd = xcc_id();
  if (threadIdx.x == 0) atomicAdd(s.census + xcd, 1);
  constexpr int EV_PER_XCD = E / NXCD;
  const int total = EV_PER_XCD * TICKETS_PER_EVENT;
  while (true) {
    if (threadIdx.x == 0) sh_ticket = atomicAdd(s.ticket + xcd * 32, 1);
    __syncthreads();
    const int ticket = sh_ticket;
    __syncthreads();
    if (ticket >= total) return;
    // tickets of a GROUP of `par` events: their par x 16 producers first, then the consumers capsule by capsule
    const int per_group = b.par * TICKETS_PER_EVENT;
    const int gi = ticket / per_group, j = ticket % per_group;
    int ei, jj;                                                    // event inside this XCD's list, ticket inside the event
    if (j < b.par * TILES) { ei = gi * b.par + j / TILES; jj = j % TILES; }
    else { const int q = j - b.par * TILES; ei = gi * b.par + (q / CONS_PER_STEP) % b.par; jj = TILES + (q / (CONS_PER_STEP * b.par)) * CONS_PER_STEP + q % CONS_PER_STEP; }
    const int e = xcd + NXCD * ei;
    float4 *ring = b.y + ((size_t)(xcd * 2 + (ei & 1)) * b.r) * K * BLK4;
    if (jj < TILES) {                                              // ---- producer of (e, tile), all capsules
      const int t = jj;
      float4 hv[P];
      const float4 *src0 = b.h + ((size_t)(e * C) * P) * BLK4 + t * 256 + tid;
#pragma unroll
      for (int p = 0; p < P; ++p) hv[p] = src0[(size_t)p * BLK4];
      for (int c = 0; c < C; ++c) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < P; ++p) { acc.x += hv[p].x; acc.y += hv[p].y; acc.z += hv[p].z; acc.w += hv[p].w; }
        const float4 *src = b.h + ((size_t)(e * C + min(c + 1, C - 1)) * P) * BLK4 + t * 256 + tid;   // the next capsule's, requested now
#pragma unroll
        for (int p = 0; p < P; ++p) hv[p] = src[(size_t)p * BLK4];
        if (c >= b.r) {                                            // the ring slot has been read by all its consumers
          if (threadIdx.x == 0) wait_for(s.taken + e * C + (c - b.r), CONS_PER_STEP, s, 0, b.sleep);
          __syncthreads();
        }
        float4 *dst = ring + ((size_t)(c % b.r) * K + sub * KT) * BLK4 + t * 256 + tid;
#pragma unroll
        for (int k = 0; k < KT; ++k) dst[(size_t)k * BLK4] = make_float4(acc.x + k, acc.y, acc.z, acc.w);   // plain: the line stays in L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(s.full + e * C + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {                                                       // ---- consumer of (e, c, two blocks)
      const int c = (jj - TILES) / CONS_PER_STEP, k = ((jj - TILES) % CONS_PER_STEP) * 2 + sub;
      if (threadIdx.x == 0) wait_for(s.full + e * C + c, TILES, s, 1, b.sleep);
      __syncthreads();
      const size_t src = ((size_t)(c % b.r) * K + k) * BLK4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = PLAIN_LOADS ? ring[src + tid + 256 * i] : ld_sc1(ring, src + tid + 256 * i);
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
      __syncthreads();                                             // every load of both blocks has landed (acc depends on them)
      if (threadIdx.x == 0) __hip_atomic_fetch_add(s.taken + e * C + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float4 *dst = b.x + ((size_t)(e * C + c) * K + k) * 2048;
#pragma unroll
      for (int i = 0; i < 8; ++i) st_nt(dst + tid + 256 * i, make_float4(acc.x + i, acc.y, acc.z, acc.w));
    }
  }
}

// ---------------------------------------------------------------- mode K: units are (event, k-tile) halves
// Same hand-off, but a unit is ONE k-tile of an event (12 blocks): 8 producer workgroups (two bin tiles each) + 6 consumer jobs per
// capsule step, 768 KB of Y per step.  H is read once per k-tile, i.e. twice per event (the price), the ring of a unit is half as
// large, so TWO OR MORE units fit an XCD's L2 at a time and their latency chains run side by side.
constexpr int UNIT_PROD = TILES / 2, UNIT_CONS = KT / 2, TICKETS_PER_UNIT = UNIT_PROD + C * UNIT_CONS;
__global__ __launch_bounds__(512, 2) void k_xcd_ks(Bufs b, Sync s) {
  __shared__ int sh_ticket;
  const int tid = threadIdx.x & 255, sub = threadIdx.x >> 8;
  const int xcd = xcc_id();
  if (threadIdx.x == 0) atomicAdd(s.census + xcd, 1);
  constexpr int UNITS_PER_XCD = 2 * E / NXCD;
  const int total = UNITS_PER_XCD * TICKETS_PER_UNIT;
  while (true) {
    if (threadIdx.x == 0) sh_ticket = atomicAdd(s.ticket + xcd * 32, 1);
    __syncthreads();
    const int ticket = sh_ticket;
    __syncthreads();
    if (ticket >= total) return;
    const int per_group = b.par * TICKETS_PER_UNIT;
    const int gi = ticket / per_group, j = ticket % per_group;
    int ui, jj;                                                    // unit inside this XCD's list, ticket inside the unit
    if (j < b.par * UNIT_PROD) { ui = gi * b.par + j / UNIT_PROD; jj = j % UNIT_PROD; }
    else { const int q = j - b.par * UNIT_PROD; ui = gi * b.par + (q / UNIT_CONS) % b.par; jj = UNIT_PROD + (q / (UNIT_CONS * b.par)) * UNIT_CONS + q % UNIT_CONS; }
    if (ui >= UNITS_PER_XCD) continue;
    const int e = xcd + NXCD * (ui >> 1), kt = ui & 1;
    int *full = s.full + (e * 2 + kt) * C, *taken = s.taken + (e * 2 + kt) * C;
    float4 *ring = b.y + ((size_t)(xcd * 8 + (ui & 7)) * b.r) * KT * BLK4;     // up to 8 units' rings per XCD
    if (jj < UNIT_PROD) {                                          // ---- producer of (e, k-tile, bin tiles 2 jj + sub), all capsules
      const int t = 2 * jj + sub;
      float4 hv[P];
      const float4 *src0 = b.h + ((size_t)(e * C) * P) * BLK4 + t * 256 + tid;
#pragma unroll
      for (int p = 0; p < P; ++p) hv[p] = src0[(size_t)p * BLK4];
      for (int c = 0; c < C; ++c) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < P; ++p) { acc.x += hv[p].x; acc.y += hv[p].y; acc.z += hv[p].z; acc.w += hv[p].w; }
        const float4 *src = b.h + ((size_t)(e * C + min(c + 1, C - 1)) * P) * BLK4 + t * 256 + tid;
#pragma unroll
        for (int p = 0; p < P; ++p) hv[p] = src[(size_t)p * BLK4];
        if (c >= b.r) {
          if (threadIdx.x == 0) wait_for(taken + (c - b.r), UNIT_CONS, s, 0, b.sleep);
          __syncthreads();
        }
        float4 *dst = ring + ((size_t)(c % b.r) * KT) * BLK4 + t * 256 + tid;
#pragma unroll
        for (int k = 0; k < KT; ++k) dst[(size_t)k * BLK4] = make_float4(acc.x + k, acc.y, acc.z, acc.w);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(full + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {                                                       // ---- consumer of (e, k-tile, c, two blocks)
      const int c = (jj - UNIT_PROD) / UNIT_CONS, k = ((jj - UNIT_PROD) % UNIT_CONS) * 2 + sub;
      if (threadIdx.x == 0) wait_for(full + c, UNIT_PROD, s, 1, b.sleep);
      __syncthreads();
      const size_t src = ((size_t)(c % b.r) * KT + k) * BLK4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = ld_sc1(ring, src + tid + 256 * i);
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(taken + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float4 *dst = b.x + ((size_t)(e * C + c) * K + kt * KT + k) * 2048;
#pragma unroll
      for (int i = 0; i < 8; ++i) st_nt(dst + tid + 256 * i, make_float4(acc.x + i, acc.y, acc.z, acc.w));
    }
  }
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("%s, %d CUs; E=%d C=%d P=%d K=%d (cfg%s regime); 512-thread workgroups\n", prop.name, n_cu, E, C, P, K, P == 6 ? "4" : "2");
  const size_t h_bytes = (size_t)E * C * P * 65536, y_full = (size_t)E * C * K * 65536, x_bytes = (size_t)E * C * K * 32768;
  float4 *h, *y, *x;
  CHECK(hipMalloc(&h, h_bytes)); CHECK(hipMalloc(&y, y_full)); CHECK(hipMalloc(&x, x_bytes));
  CHECK(hipMemset(h, 0, h_bytes)); CHECK(hipMemset(y, 0, y_full)); CHECK(hipMemset(x, 0, x_bytes));
  const int n_ints = NXCD * 32 + 4 * E * C + 1 + NXCD;
  int *ints; long long *spins;
  CHECK(hipMalloc(&ints, sizeof(int) * n_ints)); CHECK(hipMalloc(&spins, 16));
  Sync s{ints, ints + NXCD * 32, ints + NXCD * 32 + 2 * E * C, ints + NXCD * 32 + 4 * E * C, spins, ints + NXCD * 32 + 4 * E * C + 1};
  hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  const double gb_m = (h_bytes + y_full) / 1e9, gb_s = (y_full + x_bytes) / 1e9;
  printf("bytes per pass: accumulate %.2f GB (H %.2f + Y %.2f), synthesis %.2f GB (Y %.2f + x %.2f); without Y: %.2f GB\n", gb_m, h_bytes / 1e9,
         y_full / 1e9, gb_s, y_full / 1e9, x_bytes / 1e9, (h_bytes + x_bytes) / 1e9);
  float a_best = 1e9f;
  {  // ---- mode A
    Bufs b{h, y, x, 0, 1, 2};
    float bm = 1e9f, bs = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_m_loop, dim3(E * TILES), dim3(512), 0, 0, b);
      CHECK(hipEventRecord(e1));
      hipLaunchKernelGGL(k_s, dim3(E * C * CONS_PER_STEP), dim3(512), 0, 0, b);
      CHECK(hipEventRecord(e2));
      CHECK(hipEventSynchronize(e2));
      float m, sy, both; CHECK(hipEventElapsedTime(&m, e0, e1)); CHECK(hipEventElapsedTime(&sy, e1, e2)); CHECK(hipEventElapsedTime(&both, e0, e2));
      if (m < bm) bm = m; if (sy < bs) bs = sy; if (both < a_best) a_best = both;
    }
    printf("mode A (two launches, HBM-sized Y):  accumulate %.3f ms (%.2f TB/s)  synthesis %.3f ms (%.2f TB/s)  both %.3f ms\n", bm, gb_m / bm, bs,
           gb_s / bs, a_best);
  }
  // (events in flight per XCD, workgroups per CU): one event = 16 producers + consumers on the other 16 CUs' workgroups; two events
  // need 2 workgroups per CU (32 producers per XCD would leave no consumer at 1)
  struct V { int par, wpc; };
  const bool skip_x = argc > 1 && !strcmp(argv[1], "k");
  for (const V v : {V{1, 1}, V{1, 2}, V{2, 2}})
    for (int sleep : {2, 8, 32})
      for (int r : {1, 2, 3, 4, 8}) {
        if (skip_x || (size_t)NXCD * 2 * r * K * 65536 > y_full) continue;
        Bufs b{h, y, x, r, v.par, sleep};
        float best = 1e9f; int stuck = 0; long long spin[2] = {0, 0}; int census[NXCD];
        for (int rep = 0; rep < 5; ++rep) {
          CHECK(hipMemsetAsync(ints, 0, sizeof(int) * n_ints, 0)); CHECK(hipMemsetAsync(spins, 0, 16, 0));
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(k_xcd<false>, dim3(n_cu * v.wpc), dim3(512), 0, 0, b, s);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CHECK(hipMemcpy(&stuck, s.stuck, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(spin, spins, 16, hipMemcpyDeviceToHost));
          CHECK(hipMemcpy(census, s.census, sizeof(census), hipMemcpyDeviceToHost));
          if (stuck) break;
        }
        int cmin = 1 << 30, cmax = 0; for (int c : census) { cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax; }
        printf("mode X  %d event(s) in flight per XCD  %d wg/CU  poll sleep %2d  ring %d steps (%4.1f MB per XCD in flight): %.3f ms = %.2fx mode A  spins prod %lld cons %lld  wg/XCD %d..%d%s\n",
               v.par, v.wpc, sleep, r, v.par * r * K * 65536.0 / 1e6, best, best / a_best, spin[0], spin[1], cmin, cmax,
               stuck ? "  STUCK (bounded wait ran out)" : "");
        fflush(stdout);
      }
  // mode K: (event, k-tile) units; workgroups per CU 2 (launch bounds), units in flight per XCD 1..4
  for (int par : {1, 2, 3, 4})
    for (int sleep : {8, 32})
      for (int r : {2, 3, 4}) {
        Bufs b{h, y, x, r, par, sleep};
        float best = 1e9f; int stuck = 0; long long spin[2] = {0, 0}; int census[NXCD];
        for (int rep = 0; rep < 5; ++rep) {
          CHECK(hipMemsetAsync(ints, 0, sizeof(int) * n_ints, 0)); CHECK(hipMemsetAsync(spins, 0, 16, 0));
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(k_xcd_ks, dim3(n_cu * 2), dim3(512), 0, 0, b, s);
          CHECK(hipEventRecord(e1));
          CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CHECK(hipMemcpy(&stuck, s.stuck, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(spin, spins, 16, hipMemcpyDeviceToHost));
          CHECK(hipMemcpy(census, s.census, sizeof(census), hipMemcpyDeviceToHost));
          if (stuck) break;
        }
        int cmin = 1 << 30, cmax = 0; for (int c : census) { cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax; }
        printf("mode K  %d (event, k-tile) unit(s) in flight per XCD  2 wg/CU  poll sleep %2d  ring %d steps (%4.1f MB per XCD in flight; H read twice: %.2f GB): %.3f ms = %.2fx mode A  spins prod %lld cons %lld  wg/XCD %d..%d%s\n",
               par, sleep, r, par * r * KT * 65536.0 / 1e6, 2 * h_bytes / 1e9, best, best / a_best, spin[0], spin[1], cmin, cmax,
               stuck ? "  STUCK (bounded wait ran out)" : "");
        fflush(stdout);
      }
  return 0;
}
