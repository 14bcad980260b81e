// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets g++ compile audiblelight_amd/csrc/*.hip for the
// host so the kernels' index math can be checked (and run under ASan/UBSan) without a GPU.
// Each HIP thread is a coroutine (or, for sanitizer builds, a real host thread), __syncthreads() a barrier among them, wave64
// shuffles an exchange through a per-wave buffer.  One workgroup runs at a time (so `__shared__` = static).
// Never shipped, never loaded by the audiblelight_amd package: the product path is the gfx950
// build only (tests/test_hostemu_kernels.py is the sole user).
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

typedef void *hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
static const hipError_t hipErrorInvalidValue = 1;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hostemu"; }

// Two back ends with the same semantics:
//  * fibers (default): the HIP threads of a launch are ucontext coroutines on ONE host thread, scheduled round robin; a
//    barrier is "yield until everybody of my group has arrived".  No futex, no kernel entry: barrier-heavy kernels (every FFT
//    pass) run two orders of magnitude faster than with one host thread per HIP thread, which is what keeps the CPU test
//    suite at a few minutes.
//  * -DHOSTEMU_THREADS: one host thread per HIP thread and pthread barriers (sanitizer builds: ASan does not follow
//    swapcontext without annotations).
namespace hostemu {
#if defined(HOSTEMU_THREADS)
struct BlockCtx {
  pthread_barrier_t block_bar;
  std::vector<pthread_barrier_t> wave_bar;
  std::vector<double> xbuf;  // one 8-byte slot per thread for shuffles
};
inline thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
inline thread_local BlockCtx *t_ctx = nullptr;
inline void block_barrier() { pthread_barrier_wait(&t_ctx->block_bar); }
inline void wave_barrier(unsigned wave) { pthread_barrier_wait(&t_ctx->wave_bar[wave]); }

template <class F>
void launch(dim3 grid, dim3 block, F &&body) {
  const unsigned nt = block.x * block.y * block.z;
  BlockCtx ctx;
  pthread_barrier_init(&ctx.block_bar, nullptr, nt);
  const unsigned nw = (nt + 63) / 64;
  ctx.wave_bar.resize(nw);
  for (unsigned w = 0; w < nw; ++w) pthread_barrier_init(&ctx.wave_bar[w], nullptr, std::min(64u, nt - 64 * w));
  ctx.xbuf.assign(nt, 0.0);
  // one host thread per HIP thread for the whole launch; workgroups run one after another (a barrier
  // separates them because `__shared__` is a single static image).  Kernels here only return early
  // for a whole workgroup, so every thread reaches the end-of-workgroup barrier.
  std::vector<std::thread> th;
  th.reserve(nt);
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      t_blockDim = block;
      t_gridDim = grid;
      t_ctx = &ctx;
      for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
          for (unsigned bx = 0; bx < grid.x; ++bx) {
            t_blockIdx = dim3(bx, by, bz);
            body();
            if (nt > 1) pthread_barrier_wait(&ctx.block_bar);
          }
    });
  for (auto &x : th) x.join();
  pthread_barrier_destroy(&ctx.block_bar);
  for (auto &b : ctx.wave_bar) pthread_barrier_destroy(&b);
}
#else
struct Barrier {
  unsigned count = 0, members = 0, generation = 0;
};
struct BlockCtx {
  Barrier block_bar;
  std::vector<Barrier> wave_bar;
  std::vector<double> xbuf;  // one 8-byte slot per thread for shuffles
};
struct Fiber {
  ucontext_t ctx;
  void *stack = nullptr;
  bool done = false;
  dim3 threadIdx, blockIdx;
};
// the launch in progress (one at a time: the library is driven from one Python thread; calls from several are serialised)
struct Launch {
  ucontext_t scheduler;
  std::vector<Fiber> fibers;
  unsigned current = 0;
  BlockCtx ctx;
  dim3 grid, block;
  std::function<void()> body;
};
inline Launch *g_launch = nullptr;
inline std::mutex &launch_mutex() { static std::mutex m; return m; }
inline dim3 t_blockDim, t_gridDim;
#define HOSTEMU_CUR (hostemu::g_launch->fibers[hostemu::g_launch->current])
inline BlockCtx *t_ctx = nullptr;

inline void yield() { swapcontext(&HOSTEMU_CUR.ctx, &g_launch->scheduler); }
inline void barrier_wait(Barrier &b) {
  const unsigned gen = b.generation;
  if (++b.count == b.members) {
    b.count = 0;
    ++b.generation;
    return;                    // the last one to arrive goes straight on
  }
  while (b.generation == gen) yield();
}
inline void block_barrier() { barrier_wait(t_ctx->block_bar); }
inline void wave_barrier(unsigned wave) { barrier_wait(t_ctx->wave_bar[wave]); }

inline void fiber_entry() {
  Launch &L = *g_launch;
  Fiber &me = L.fibers[L.current];
  const unsigned nt = (unsigned)L.fibers.size();
  for (unsigned bz = 0; bz < L.grid.z; ++bz)
    for (unsigned by = 0; by < L.grid.y; ++by)
      for (unsigned bx = 0; bx < L.grid.x; ++bx) {
        me.blockIdx = dim3(bx, by, bz);
        L.body();
        if (nt > 1) block_barrier();   // `__shared__` is one static image: the next workgroup starts when this one is done
      }
  me.done = true;
  swapcontext(&me.ctx, &L.scheduler);
}

template <class F>
void launch(dim3 grid, dim3 block, F &&body) {
  std::lock_guard<std::mutex> guard(launch_mutex());
  const unsigned nt = block.x * block.y * block.z;
  constexpr size_t STACK = 512 * 1024;
  static std::vector<void *> stacks;   // reused between launches
  while (stacks.size() < nt) stacks.push_back(mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0));
  Launch L;
  L.grid = grid;
  L.block = block;
  L.body = [&] { body(); };
  L.ctx.block_bar.members = nt;
  const unsigned nw = (nt + 63) / 64;
  L.ctx.wave_bar.resize(nw);
  for (unsigned w = 0; w < nw; ++w) L.ctx.wave_bar[w].members = std::min(64u, nt - 64 * w);
  L.ctx.xbuf.assign(nt, 0.0);
  L.fibers.resize(nt);
  g_launch = &L;
  t_ctx = &L.ctx;
  t_blockDim = block;
  t_gridDim = grid;
  for (unsigned t = 0; t < nt; ++t) {
    Fiber &f = L.fibers[t];
    f.threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = stacks[t];
    f.ctx.uc_stack.ss_size = STACK;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, fiber_entry, 0);
  }
  for (unsigned alive = nt; alive > 0;) {       // round robin: a fiber runs until its next barrier (or its end)
    for (unsigned t = 0; t < nt; ++t) {
      if (L.fibers[t].done) continue;
      L.current = t;
      swapcontext(&L.scheduler, &L.fibers[t].ctx);
      if (L.fibers[t].done) --alive;
    }
  }
  g_launch = nullptr;
  t_ctx = nullptr;
}
#endif
}  // namespace hostemu

#if defined(HOSTEMU_THREADS)
#define threadIdx (hostemu::t_threadIdx)
#define blockIdx (hostemu::t_blockIdx)
#else
#define threadIdx (HOSTEMU_CUR.threadIdx)
#define blockIdx (HOSTEMU_CUR.blockIdx)
#endif
#define blockDim (hostemu::t_blockDim)
#define gridDim (hostemu::t_gridDim)

static inline void __syncthreads() { hostemu::block_barrier(); }

template <class T>
static inline T __shfl_down(T v, unsigned off, int /*width*/ = 64) {
  static_assert(sizeof(T) <= 8, "shuffle slot is 8 bytes");
  auto *c = hostemu::t_ctx;
  const unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  const unsigned lane = tid & 63, wave = tid >> 6;
  memcpy(&c->xbuf[tid], &v, sizeof(T));
  hostemu::wave_barrier(wave);
  T r = v;
  if (lane + off < 64 && tid + off < c->xbuf.size()) memcpy(&r, &c->xbuf[tid + off], sizeof(T));
  hostemu::wave_barrier(wave);
  return r;
}

static inline void sincospi(double x, double *s, double *c) {
  *s = sin(M_PI * x);
  *c = cos(M_PI * x);
}
static inline float sinpif(float x) { return (float)sin(M_PI * (double)x); }
static inline void sincospif(float x, float *s, float *c) {
  *s = (float)sin(M_PI * (double)x);
  *c = (float)cos(M_PI * (double)x);
}
static inline unsigned __float_as_uint(float x) { unsigned u; memcpy(&u, &x, 4); return u; }
using std::isfinite;
using std::max;
using std::min;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hostemu::launch((grid), (block), [&] { kernel(__VA_ARGS__); })
