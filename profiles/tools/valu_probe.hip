// Issue rate of scalar vs packed f32 VALU instructions on gfx950.  Measurement tool only.
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

// MODE 0: v_fma_f32 x16 chains; 1: v_pk_fma_f32 x8 chains (same flops per iteration as 16 scalar);
// 2: v_pk_add_f32; 3: v_pk_mul_f32; 4: v_add_f32; 5: v_pk_fma_f32 with op_sel/neg modifiers (complex MAC form)
template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float *sink, int iters) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
  float b0 = 1.0001f, b1 = 0.9999f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i].x = a[2 * i]; p[i].y = a[2 * i + 1]; }
  v2 q; q.x = b0; q.y = b1;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b0), "v"(b1));
    } else if constexpr (MODE == 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b1));
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(q));
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q));
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_pk_fma_f32 %0, %1, %1, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(p[i]) : "v"(q));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  if (s == 12345.678f) sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *sink, int waves_per_simd, int instr_per_iter, int flops_per_instr_lane) {
  const int iters = 20000;
  dim3 grid(256 * waves_per_simd), block(256);  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_valu<MODE>), grid, block, 0, 0, sink, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_valu<MODE>), grid, block, 0, 0, sink, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)iters * instr_per_iter * waves_per_simd;
  double ns_per_instr = ms * 1e6 / instr_per_simd;
  double tflops = 1024.0 * instr_per_simd * 64 * flops_per_instr_lane / (ms * 1e-3) / 1e12;
  printf("%-28s %d waves/SIMD  %.3f ms  %.3f ns per wave-instruction per SIMD (%.2f cyc @2.4GHz)  %.1f TFLOP/s chip\n", name,
         waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4, tflops);
}

int main() {
  float *sink; hipMalloc(&sink, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<4>("v_add_f32", sink, w, 64, 1);
    run<0>("v_fma_f32", sink, w, 64, 2);
    run<2>("v_pk_add_f32", sink, w, 32, 2);
    run<3>("v_pk_mul_f32", sink, w, 32, 2);
    run<1>("v_pk_fma_f32", sink, w, 32, 4);
    run<5>("v_pk_fma_f32 op_sel/neg", sink, w, 32, 4);
  }
  return 0;
}
