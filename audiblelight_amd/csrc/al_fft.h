// In-LDS Stockham FFT for gfx950 (CDNA4): the per-block transform of the partitioned
// overlap-save convolution (DESIGN.md "Kernels").
//
// Geometry (FftGeom<LOG2M, E>): a transform of M = 2^LOG2M complex points is done by one workgroup of
// T = M/E threads, every thread owning E complex values (E = 16 or 32), always the LDS slots
// tid + T*m, m < E, on the read side (contiguous 512 bytes per wave: conflict-free ds_read_b64), and
// scattering its butterfly outputs in Stockham auto-sort order on the write side.  The first pass is a
// radix-E butterfly entirely in registers; the remaining log2(M/E) bits are done in passes of radix
// <= E with E/R butterflies per thread.  A pass is: read E -> twiddle -> butterflies -> barrier ->
// write E -> barrier, in place (all reads of a pass complete before any write).
//
// Why two widths: what the transform waits on is the LDS exchange between passes and its two workgroup
// barriers (profiles/r01_fft_probe2.txt: M = 8192 with 16 values per thread needs four passes, the last
// a bare radix 2, and runs at 60 % of the VALU issue rate; M = 4096, three passes, reaches 85 %).  With 32
// values per thread M = 8192 is 32 x 16 x 16: three passes, two exchanges, and half as many waves at
// every barrier.
//
// The LDS image is padded by one float2 per E so the stride-E scatter of the first pass lands on
// distinct banks (ds_write_b64 is serviced 16 lanes at a time over 32 dword banks: row pitch E+1
// complex = 2 dwords mod 32).
//
// Twiddles: one table tw[k] = exp(-i*pi*k/M), k < M (the 2M-th roots over half a turn, made in float64
// by al_twiddle_init).  A thread loads its whole set once (FftTwiddles): (w, w^4[, w^16]) per butterfly of each
// pass, from which the other powers follow by at most three chained complex products, and tw[tid], from
// which the real-packing steps derive their factors by one product with a compile-time constant.
#pragma once
#include <hip/hip_runtime.h>

#include "al_common.h"

namespace al {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// a += b * c
__device__ __forceinline__ void cfma(float2 &a, float2 b, float2 c) {
  a.x = fmaf(b.x, c.x, a.x);
  a.x = fmaf(-b.y, c.y, a.x);
  a.y = fmaf(b.x, c.y, a.y);
  a.y = fmaf(b.y, c.x, a.y);
}
// multiply by DIR*i  (DIR = -1 forward transform, +1 inverse)
template <int DIR>
__device__ __forceinline__ float2 rot90(float2 a) {
  return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

// Compile-time sine / cosine of pi*num/den (0 <= num/den <= 1/2) for the constant factors of the packing steps.
constexpr double ct_sin_series(double x) {  // |x| <= pi/4
  double term = x, sum = x;
  for (int i = 1; i < 12; ++i) {
    term *= -x * x / ((2 * i) * (2 * i + 1));
    sum += term;
  }
  return sum;
}
constexpr double ct_cos_series(double x) {  // |x| <= pi/4
  double term = 1.0, sum = 1.0;
  for (int i = 1; i < 12; ++i) {
    term *= -x * x / ((2 * i - 1) * (2 * i));
    sum += term;
  }
  return sum;
}
constexpr double CT_PI = 3.14159265358979323846;
constexpr double ct_sinpi(int num, int den) {
  return 4 * num <= den ? ct_sin_series(CT_PI * num / den) : ct_cos_series(CT_PI * (den - 2 * num) / (2.0 * den));
}
constexpr double ct_cospi(int num, int den) {
  return 4 * num <= den ? ct_cos_series(CT_PI * num / den) : ct_sin_series(CT_PI * (den - 2 * num) / (2.0 * den));
}

// Values per thread of the product kernels: 32 from 8192 points up (see the header), 16 below.
constexpr int fft_default_elems(int log2m) { return log2m >= 13 ? 32 : 16; }

template <int LOG2M, int ELEMS>
struct FftGeom {
  static_assert(ELEMS == 16 || ELEMS == 32, "16 or 32 complex values per thread");
  static constexpr int E = ELEMS;
  static constexpr int M = 1 << LOG2M;
  static constexpr int LOG2E = E == 32 ? 5 : 4;
  static constexpr int T = M / E;  // threads per workgroup
  static constexpr int H = E / 2;
  static constexpr int REST = LOG2M - LOG2E;
  static constexpr int NPASSES = 1 + (E == 16 ? (REST + 3) / 4 : (REST + 4) / 5);
  static_assert(T % E == 0 && T >= 64, "slot offsets T*m must be multiples of the padding period");
  static_assert(NPASSES <= (E == 16 ? 4 : 3), "pass plan covers LOG2M <= 16 (E = 16) / 15 (E = 32)");
  static constexpr int pad(int i) { return i + (i >> LOG2E); }  // pad(a + c) == pad(a) + pad(c) when c % E == 0
  static constexpr int LDS_ELEMS = pad(M);
  // bits resolved by each pass: the first is log2 E; E = 16 then takes 4 at a time with the remainder last,
  // E = 32 splits what is left over at most two passes, the wider one first
  static constexpr int bits(int pass) {
    if (pass >= NPASSES) return 0;
    if (pass == 0) return LOG2E;
    if (E == 16) return (REST - 4 * (pass - 1)) >= 4 ? 4 : (REST - 4 * (pass - 1));
    if (NPASSES == 2) return REST;
    return pass == 1 ? (REST + 1) / 2 : REST / 2;
  }
  static constexpr int radix(int pass) { return 1 << bits(pass); }
  static constexpr int ns(int pass) {  // product of the radices before `pass`
    int n = 1;
    for (int p = 0; p < pass; ++p) n *= radix(p);
    return n;
  }
};

template <int DIR>
__device__ __forceinline__ void bfly2(float2 &a, float2 &b) {
  float2 t = csub(a, b);
  a = cadd(a, b);
  b = t;
}

template <int DIR>
__device__ __forceinline__ void bfly4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
  float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = rot90<DIR>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

// x[0..7] natural order in, X[0..7] natural order out.
template <int DIR, int N>
__device__ __forceinline__ void bfly8(float2 (&x)[N]) {
  constexpr float H = 0.70710678118654752440f;
  bfly4<DIR>(x[0], x[2], x[4], x[6]);  // even samples  -> A0[k1] at x[0],x[2],x[4],x[6]
  bfly4<DIR>(x[1], x[3], x[5], x[7]);  // odd samples   -> A1[k1] at x[1],x[3],x[5],x[7]
  // A1[k1] *= w8^(k1), w8 = exp(DIR * 2*pi*i/8)
  float2 b1 = x[3], b3 = x[7];
  x[3] = make_float2(H * (b1.x - DIR * b1.y), H * (b1.y + DIR * b1.x));
  x[5] = rot90<DIR>(x[5]);
  x[7] = make_float2(H * (-b3.x - DIR * b3.y), H * (-b3.y + DIR * b3.x));
  float2 a0 = x[0], a1 = x[2], a2 = x[4], a3 = x[6];
  float2 c0 = x[1], c1 = x[3], c2 = x[5], c3 = x[7];
  x[0] = cadd(a0, c0); x[4] = csub(a0, c0);
  x[1] = cadd(a1, c1); x[5] = csub(a1, c1);
  x[2] = cadd(a2, c2); x[6] = csub(a2, c2);
  x[3] = cadd(a3, c3); x[7] = csub(a3, c3);
}

// 16 points x[OFF..OFF+15] (natural order) of a register array; X[k] ends up at x[OFF + 4*(k%4) + k/4].
template <int DIR, int OFF, int N>
__device__ __forceinline__ void bfly16(float2 (&x)[N]) {
  constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, H = 0.70710678118654752440f;
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) bfly4<DIR>(x[OFF + n2], x[OFF + 4 + n2], x[OFF + 8 + n2], x[OFF + 12 + n2]);
  // now x[4*k1 + n2] = A[n2][k1]; multiply by w16^(n2*k1), w16 = exp(DIR*2*pi*i/16)
  const float2 w1 = make_float2(C1, DIR * S1), w2 = make_float2(H, DIR * H), w3 = make_float2(S1, DIR * C1);
  const float2 w6 = make_float2(-H, DIR * H), w9 = make_float2(-C1, -DIR * S1);
  x[OFF + 5] = cmul(x[OFF + 5], w1);          // k1=1,n2=1
  x[OFF + 6] = cmul(x[OFF + 6], w2);          // k1=1,n2=2
  x[OFF + 7] = cmul(x[OFF + 7], w3);          // k1=1,n2=3
  x[OFF + 9] = cmul(x[OFF + 9], w2);          // k1=2,n2=1
  x[OFF + 10] = rot90<DIR>(x[OFF + 10]);      // k1=2,n2=2 : w4
  x[OFF + 11] = cmul(x[OFF + 11], w6);        // k1=2,n2=3
  x[OFF + 13] = cmul(x[OFF + 13], w3);        // k1=3,n2=1
  x[OFF + 14] = cmul(x[OFF + 14], w6);        // k1=3,n2=2
  x[OFF + 15] = cmul(x[OFF + 15], w9);        // k1=3,n2=3
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) bfly4<DIR>(x[OFF + 4 * k1], x[OFF + 4 * k1 + 1], x[OFF + 4 * k1 + 2], x[OFF + 4 * k1 + 3]);
}

// 32 points, natural order in.  Decimation in frequency by 2, then two 16-point butterflies:
//   X[2 q2]     = DFT16( x[r2] + x[16 + r2] )[q2]               -> x[     pos16(q2)]
//   X[2 q2 + 1] = DFT16( (x[r2] - x[16 + r2]) * w32^r2 )[q2]    -> x[16 + pos16(q2)]
template <int DIR>
__device__ __forceinline__ void bfly32(float2 (&x)[32]) {
  constexpr float C[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                           0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f,
                           0.f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
                           -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
  constexpr float S[16] = {0.f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                           0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f,
                           1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                           0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float2 a = x[r], b = x[16 + r];
    x[r] = cadd(a, b);
    const float2 d = csub(a, b);
    if (r == 0) x[16] = d;
    else if (r == 8) x[24] = rot90<DIR>(d);
    else x[16 + r] = cmul(d, make_float2(C[r], DIR * S[r]));
  }
  bfly16<DIR, 0>(x);
  bfly16<DIR, 16>(x);
}

// Where output q of the R-point butterfly sits after the in-place routine above.
constexpr int bfly_pos(int R, int q) {
  return R == 16 ? 4 * (q & 3) + (q >> 2) : R == 32 ? 16 * (q & 1) + 4 * ((q >> 1) & 3) + (q >> 3) : q;
}

template <int DIR, int R>
__device__ __forceinline__ void butterfly(float2 (&x)[R]) {
  if constexpr (R == 32) bfly32<DIR>(x);
  else if constexpr (R == 16) bfly16<DIR, 0>(x);
  else if constexpr (R == 8) bfly8<DIR>(x);
  else if constexpr (R == 4) bfly4<DIR>(x[0], x[1], x[2], x[3]);
  else bfly2<DIR>(x[0], x[1]);
}

// Twiddle powers w^1..w^(R-1) from exact table entries (w^1, w^4, w^16): at most three chained products
// per power.
template <int R>
__device__ __forceinline__ void twiddle_powers(float2 (&w)[R], float2 a, float2 b, float2 d) {
  w[1] = a;
  if constexpr (R > 2) {
    w[2] = cmul(a, a);
    w[3] = cmul(w[2], a);
  }
  if constexpr (R > 4) {
    w[4] = b;
    w[5] = cmul(b, a);
    w[6] = cmul(b, w[2]);
    w[7] = cmul(b, w[3]);
  }
  if constexpr (R > 8) {
    w[8] = cmul(b, b);
    w[9] = cmul(w[8], a);
    w[10] = cmul(w[8], w[2]);
    w[11] = cmul(w[8], w[3]);
    w[12] = cmul(w[8], b);
    w[13] = cmul(w[12], a);
    w[14] = cmul(w[12], w[2]);
    w[15] = cmul(w[12], w[3]);
  }
  if constexpr (R > 16) {
    w[16] = d;
#pragma unroll
    for (int i = 1; i < 16; ++i) w[16 + i] = cmul(d, w[i]);
  }
}

// Twiddle base factors of one pass: one exact (w^1, w^4, w^16) triple per butterfly.
template <class G, int PASS>
struct PassTwiddles {
  static constexpr int NB = G::radix(PASS) > 1 ? G::E / G::radix(PASS) : 1;
  float2 a[NB], c[NB], d[NB];
};

// TS: the table was made for TS * G::M points (tw[k] = exp(-i*pi*k / (TS*M))); every index is scaled by TS.
template <class G, int DIR, int PASS, int TS = 1>
__device__ __forceinline__ void load_pass_twiddles(PassTwiddles<G, PASS> &t, const float2 *__restrict__ tw, int tid) {
  if constexpr (PASS > 0 && PASS < G::NPASSES) {
    constexpr int R = G::radix(PASS), NS = G::ns(PASS), NB = G::E / R;
    constexpr int STEP = TS * 2 * G::M / (NS * R);  // tw[k*STEP] = exp(-2*pi*i*k/(NS*R))
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int k = (tid + G::T * b) & (NS - 1);
      t.a[b] = tw[k * STEP];
      t.c[b] = (R > 4) ? tw[4 * k * STEP] : t.a[b];
      t.d[b] = (R > 16) ? tw[16 * k * STEP] : t.a[b];
      if (DIR > 0) {
        t.a[b].y = -t.a[b].y;
        t.c[b].y = -t.c[b].y;
        t.d[b].y = -t.d[b].y;
      }
    }
  }
}

// Returns x unchanged but unknown to the optimiser (no instruction is emitted).
__device__ __forceinline__ void make_opaque(float2 &x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x.x), "+v"(x.y));
#endif
}

// Every table value one transform of this thread needs: the pass factors and tw[tid], from which the packing steps
// derive their factors (below).  They depend on the lane only, so a workgroup that transforms several blocks
// loads them once; and because they are requested before anything else, no later wait on them can be held up
// behind a data prefetch (vector-memory results are counted in order).
template <class G>
struct FftTwiddles {
  PassTwiddles<G, 1> p1;
  PassTwiddles<G, 2> p2;
  PassTwiddles<G, 3> p3;
  float2 w0;  // tw[tid] = exp(-i*pi*tid/M)
  // In a loop over blocks call this at the top of the body: everything DERIVED from the factors (fifteen powers per
  // butterfly) would otherwise be hoisted out of the loop as invariant and held in registers across it.
  __device__ __forceinline__ void hide_from_hoisting() {
    hide(p1);
    hide(p2);
    hide(p3);
    make_opaque(w0);
  }
  template <int PASS>
  static __device__ __forceinline__ void hide(PassTwiddles<G, PASS> &p) {
    if constexpr (PASS < G::NPASSES) {
#pragma unroll
      for (int i = 0; i < PassTwiddles<G, PASS>::NB; ++i) {
        make_opaque(p.a[i]);
        if (G::radix(PASS) > 4) make_opaque(p.c[i]);
        if (G::radix(PASS) > 16) make_opaque(p.d[i]);
      }
    }
  }
  template <int PASS>
  __device__ __forceinline__ const PassTwiddles<G, PASS> &pass() const {
    if constexpr (PASS == 1) return p1;
    else if constexpr (PASS == 2) return p2;
    else return p3;
  }
};

template <class G, int DIR, int TS = 1>
__device__ __forceinline__ void load_fft_twiddles(FftTwiddles<G> &t, const float2 *__restrict__ tw, int tid) {
  static_assert(G::NPASSES <= 4, "FftTwiddles holds three twiddled passes");
  load_pass_twiddles<G, DIR, 1, TS>(t.p1, tw, tid);
  load_pass_twiddles<G, DIR, 2, TS>(t.p2, tw, tid);
  load_pass_twiddles<G, DIR, 3, TS>(t.p3, tw, tid);
  t.w0 = tw[tid * TS];
}

// Packing-step factors: tw[tid + T*m] = tw[tid] * exp(-i*pi*m/E) (T/M = 1/E), the second factor a constant.
template <class G>
struct PackFactors {
  float c[G::H], s[G::H];
  constexpr PackFactors() : c{}, s{} {
    for (int m = 0; m < G::H; ++m) {
      c[m] = (float)ct_cospi(m, G::E);
      s[m] = (float)ct_sinpi(m, G::E);
    }
  }
};

// One Stockham pass.  On entry (PASS == 0) v[m] holds in[tid + T*m]; later passes read those
// slots from LDS themselves.  Outputs are scattered to LDS; the caller must have a barrier
// between the last read of the LDS image and this call's writes when PASS == 0.
//   butterfly j = tid + T*b reads in[j + r*M/R]      = v[b + r*NB]
//   k = j mod NS;  factor exp(DIR*2*pi*i*r*k/(NS*R))
//   out[(j - k)*R + k + q*NS] = X[q]
template <class G, int DIR, int PASS, bool TO_REGS = false, bool LDS_ONLY = false>
__device__ __forceinline__ void fft_pass(float2 (&v)[G::E], float2 *s, const PassTwiddles<G, PASS> &t, int tid) {
  constexpr int E = G::E, T = G::T;
  constexpr int R = G::radix(PASS), NS = G::ns(PASS), NB = E / R;
  static_assert(PASS > 0 || R == E, "the first pass is one radix-E butterfly per thread");
  if (PASS > 0) {
    const float2 *rd = s + G::pad(tid);  // slot m lives at rd[pad(T*m)]: one base register, immediate offsets
#pragma unroll
    for (int m = 0; m < E; ++m) v[m] = rd[G::pad(T * m)];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float2 w[R];
      twiddle_powers<R>(w, t.a[b], t.c[b], t.d[b]);
#pragma unroll
      for (int r = 1; r < R; ++r) v[b + r * NB] = cmul(v[b + r * NB], w[r]);
    }
    // all LDS reads of this pass are done once every thread is here
    block_barrier<LDS_ONLY>();
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float2 x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = v[b + r * NB];
    butterfly<DIR, R>(x);
    // Final pass (NS * R == M): output q of butterfly b is element tid + T*(b + q*NB), i.e. the thread's own
    // slot b + q*NB, so with TO_REGS the natural-order result stays in registers: v[m] = X[tid + T*m].
#pragma unroll
    for (int q = 0; q < R; ++q) v[b + q * NB] = x[bfly_pos(R, q)];
  }
  if (TO_REGS) return;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = tid + T * b;
    const int k = j & (NS - 1);
    const int base = (j - k) * R + k;
    // output q goes to base + q*NS.  NS == 1 (first pass, R == E): pad(E*j + q) = (E+1)*j + q.
    // NS >= E: q*NS is a multiple of E, so pad() splits into a per-thread base and a constant offset.
    float2 *wr = s + (NS == 1 ? (E + 1) * j : G::pad(base));
    constexpr int QS = (NS == 1) ? 1 : G::pad(NS);  // NS >= E: pad(q*NS) == q * pad(NS)
#pragma unroll
    for (int q = 0; q < R; ++q) wr[q * QS] = v[b + q * NB];
  }
  block_barrier<LDS_ONLY>();
}

template <class G, int DIR, int PASS, bool LAST_REGS, bool LDS_ONLY = false>
struct FftPasses {
  static __device__ __forceinline__ void run(float2 (&v)[G::E], float2 *s, const FftTwiddles<G> &t, int tid) {
    if constexpr (PASS == 0) {
      PassTwiddles<G, 0> none;  // pass 0 has unit twiddles
      fft_pass<G, DIR, 0, false, LDS_ONLY>(v, s, none, tid);
      FftPasses<G, DIR, 1, LAST_REGS, LDS_ONLY>::run(v, s, t, tid);
    } else if constexpr (PASS + 1 < G::NPASSES) {
      fft_pass<G, DIR, PASS, false, LDS_ONLY>(v, s, t.template pass<PASS>(), tid);
      FftPasses<G, DIR, PASS + 1, LAST_REGS, LDS_ONLY>::run(v, s, t, tid);
    } else {
      fft_pass<G, DIR, PASS, LAST_REGS, LDS_ONLY>(v, s, t.template pass<PASS>(), tid);
    }
  }
};

// Complex FFT of the M values held as v[m] = in[tid + T*m]; the natural-order result comes back in the same
// registers: v[m] = X[tid + T*m].  `t` must have been loaded with the same DIR.  The last pass ends its LDS reads
// with a barrier, so the caller may write the LDS image immediately.
template <class G, int DIR>
__device__ __forceinline__ void fft_regs_to_regs(float2 (&v)[G::E], float2 *s, const FftTwiddles<G> &t, int tid) {
  static_assert(G::NPASSES >= 2, "single-pass transforms are not used");
  FftPasses<G, DIR, 0, true>::run(v, s, t, tid);
}

// ---- where slot q of a spectrum half lives
// PlainSlots: at q (the layouts of DESIGN.md sections 2 and 5).  The slot map is a template parameter of the packing steps.
struct PlainSlots {
  static __device__ __forceinline__ void store_even(float2 *half, int q, float2 v) { stream_store<4>(half + q, v); }
  static __device__ __forceinline__ void store_odd(float2 *half, int q, float2 v) { stream_store<4>(half + q, v); }
  static __device__ __forceinline__ float2 load_even(const float2 *half, int q) { return stream_load<2>(half + q); }
  static __device__ __forceinline__ float2 load_odd(const float2 *half, int q) { return stream_load<2>(half + q); }
};
// ---- real <-> half-complex packing around the M-point complex transform (N = 2M real samples)
// Forward: Z = FFT_M(x[2n] + i x[2n+1]);  X[k] = E + w^k O,  X[M-k] = conj(E - w^k O),
//   E = (Z[k] + conj Z[M-k])/2,  O = -i (Z[k] - conj Z[M-k])/2,  w = exp(-i*pi/M).
// Spectrum layout: out[0] = (X[0], X[M]) (both real), out[k] = X[k] for 0 < k < M.
// The transform result lives in v[m] = Z[tid + T*m]; only the mirrored half (Z[M-k], owned by thread T-tid) goes
// through LDS.
template <class G, class L = PlainSlots, bool LDS_ONLY = false>
__device__ __forceinline__ void real_unpack_store_regs(const float2 (&v)[G::E], float2 *s, float2 w0, int tid,
                                                       float2 *__restrict__ out) {
  constexpr int M = G::M, T = G::T, E = G::E, H = G::H;
  constexpr PackFactors<G> pf{};
  float2 *own = s + G::pad(tid);
#pragma unroll
  for (int m = H; m < E; ++m) own[G::pad(T * m)] = v[m];
  block_barrier<LDS_ONLY>();
#pragma unroll
  for (int m = 0; m < H; ++m) {
    const int k = tid + T * m;
    if (m == 0 && k == 0) {
      const float2 z0 = v[0], zh = v[H];  // thread 0 owns Z[0] and Z[M/2]
      out[0] = make_float2(z0.x + z0.y, z0.x - z0.y);
      L::store_even(out, M / 2, cconj(zh));
    } else {
      const float2 wm = m == 0 ? w0 : cmul(w0, make_float2(pf.c[m], -pf.s[m]));
      const float2 zk = v[m], zm = s[G::pad(M - k)];
      const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
      const float2 d = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
      const float2 o = make_float2(d.y, -d.x);
      const float2 wo = cmul(wm, o);
      L::store_even(out, k, cadd(e, wo));
      L::store_even(out, M - k, cconj(csub(e, wo)));
    }
  }
}

// Inverse: from the packed spectrum Y build Z[k] = E + iO (scaled), ready for the inverse passes.
//   E = (Y[k] + conj Y[M-k])/2, O = conj(w^k) (Y[k] - conj Y[M-k])/2.
// Split in two so the caller decides what runs while the global loads are in flight.
template <class G, class L = PlainSlots>
__device__ __forceinline__ void real_pack_issue(const float2 *__restrict__ in, float2 (&yk)[G::H], float2 (&ym)[G::H], int tid) {
#pragma unroll
  for (int m = 0; m < G::H; ++m) {
    const int k = tid + G::T * m;
    yk[m] = k == 0 ? stream_load<2>(in) : L::load_even(in, k);     // slot 0 (packed DC / Nyquist) is slot 0 in every layout
    ym[m] = L::load_even(in, k == 0 ? G::M / 2 : G::M - k);
  }
}

template <class G>
__device__ __forceinline__ void real_pack_finish(const float2 (&yk)[G::H], const float2 (&ym)[G::H], float2 (&v)[G::E],
                                                 float2 *s, float2 w0, int tid, float scale) {
  constexpr int M = G::M, T = G::T, E = G::E, H = G::H;
  constexpr PackFactors<G> pf{};
  const float hs = 0.5f * scale;
#pragma unroll
  for (int m = 0; m < H; ++m) {
    const int k = tid + T * m;
    if (k == 0) {
      v[0] = make_float2(hs * (yk[0].x + yk[0].y), hs * (yk[0].x - yk[0].y));
      s[G::pad(M / 2)] = make_float2(scale * ym[0].x, -scale * ym[0].y);
    } else {
      const float2 e = make_float2(hs * (yk[m].x + ym[m].x), hs * (yk[m].y - ym[m].y));
      const float2 d = make_float2(hs * (yk[m].x - ym[m].x), hs * (yk[m].y + ym[m].y));
      const float2 wm = m == 0 ? w0 : cmul(w0, make_float2(pf.c[m], -pf.s[m]));
      const float2 o = cmul(cconj(wm), d);
      v[m] = make_float2(e.x - o.y, e.y + o.x);
      s[G::pad(M - k)] = make_float2(e.x + o.y, o.x - e.y);
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = H; m < E; ++m) v[m] = s[G::pad(tid + T * m)];
  __syncthreads();
}

template <class G, class L = PlainSlots>
__device__ __forceinline__ void real_pack_load_regs(const float2 *__restrict__ in, float2 (&v)[G::E], float2 *s,
                                                    float2 w0, int tid, float scale) {
  float2 yk[G::H], ym[G::H];
  real_pack_issue<G, L>(in, yk, ym, tid);
  real_pack_finish<G>(yk, ym, v, s, w0, tid, scale);
}

}  // namespace al
